// Axis-angle <-> rotation matrix device functions shared by the body-model kernels (fk.hip) and the task loops (tasks.hip).
//   rodrigues      -- smplx==0.1.28 lbs.py batch_rodrigues (un-vendored dependency; what lib/utils/transforms.py:258-261 reaches through
//                     torchgeometry agrees with it to fp32 rounding)
//   rodrigues_bwd  -- its vector-Jacobian product
#pragma once
#include <hip/hip_runtime.h>

struct Mat3 {
    float m[9];
};

// sin and cos with a 3-term Cody-Waite reduction by pi/2 and the cephes minimax polynomials on [-pi/4, pi/4]
// (~1 ulp for |x| < 1e4, no slow path): libm's sincosf carries a Payne-Hanek large-argument branch that costs
// ~150 instructions per call, 3x the rest of a joint.
__device__ __forceinline__ void sincos_small(float x, float& s, float& c) {
    const float k = rintf(x * 0.636619772367581343f);
    float r = fmaf(k, -1.57079625129699707031f, x);
    r = fmaf(k, -7.54978941586159635335e-08f, r);
    r = fmaf(k, -5.39030285815811905290e-15f, r);
    const float r2 = r * r;
    float sp = fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    sp = fmaf(sp, r2, -1.6666654611e-1f);
    sp = fmaf(sp * r2, r, r);
    float cp = fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    cp = fmaf(cp, r2, 4.166664568298827e-2f);
    cp = fmaf(cp * r2, r2, fmaf(-0.5f, r2, 1.0f));
    const int q = (int)k;
    const float ss = (q & 1) ? cp : sp, cc = (q & 1) ? sp : cp;
    s = (q & 2) ? -ss : ss;
    c = ((q + 1) & 2) ? -cc : cc;
}

// smplx lbs.py batch_rodrigues: angle = ||r + 1e-8||, k = r / angle, R = I + sin K + (1 - cos) K K
__device__ __forceinline__ Mat3 rodrigues(float rx, float ry, float rz) {
    const float ax = rx + 1e-8f, ay = ry + 1e-8f, az = rz + 1e-8f;
    const float d2 = fmaf(ax, ax, fmaf(ay, ay, az * az));
    const float inv = rsqrtf(d2);                        // v_rsq_f32 (1 ulp); d2 >= 3e-16 thanks to the 1e-8 offset
    const float angle = d2 * inv;
    const float kx = rx * inv, ky = ry * inv, kz = rz * inv;
    float s, c;
    sincos_small(angle, s, c);
    const float c1 = 1.0f - c;
    Mat3 R;
    R.m[0] = 1.0f + c1 * (-(kz * kz) - ky * ky);
    R.m[1] = s * (-kz) + c1 * (kx * ky);
    R.m[2] = s * ky + c1 * (kx * kz);
    R.m[3] = s * kz + c1 * (kx * ky);
    R.m[4] = 1.0f + c1 * (-(kz * kz) - kx * kx);
    R.m[5] = s * (-kx) + c1 * (ky * kz);
    R.m[6] = s * (-ky) + c1 * (kx * kz);
    R.m[7] = s * kx + c1 * (ky * kz);
    R.m[8] = 1.0f + c1 * (-(ky * ky) - kx * kx);
    return R;
}

// Rodrigues backward: R = I + s K + c1 K^2, K = skew(k), k = r / angle;  o[3] = d loss / d (rx, ry, rz) from dR = d loss / d R
__device__ __forceinline__ void rodrigues_bwd(float rx, float ry, float rz, const float (&dR)[9], float* o) {
    const float ax = rx + 1e-8f, ay = ry + 1e-8f, az = rz + 1e-8f;
    const float angle = sqrtf(ax * ax + ay * ay + az * az);
    const float inv = 1.0f / angle;
    const float kx = rx * inv, ky = ry * inv, kz = rz * inv;
    float sn, cs;
    sincos_small(angle, sn, cs);
    const float c1 = 1.0f - cs;
    const float K[9] = {0.f, -kz, ky, kz, 0.f, -kx, -ky, kx, 0.f};
    float KK[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) KK[3 * r + c] = K[3 * r] * K[c] + K[3 * r + 1] * K[3 + c] + K[3 * r + 2] * K[6 + c];
    float ds = 0.f, dc1 = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) { ds += dR[k] * K[k]; dc1 += dR[k] * KK[k]; }
    float dK[9];                                   // s dR + c1 (dR K^T + K^T dR)
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int m = 0; m < 3; ++m) { t1 += dR[3 * r + m] * K[3 * c + m]; t2 += K[3 * m + r] * dR[3 * m + c]; }
            dK[3 * r + c] = sn * dR[3 * r + c] + c1 * (t1 + t2);
        }
    const float dk[3] = {dK[7] - dK[5], dK[2] - dK[6], dK[3] - dK[1]};
    const float dtheta = ds * cs + dc1 * sn;
    const float kdk = kx * dk[0] + ky * dk[1] + kz * dk[2];
    const float kv[3] = {kx, ky, kz};
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = (dk[c] - kdk * kv[c]) * inv + dtheta * kv[c];
}
