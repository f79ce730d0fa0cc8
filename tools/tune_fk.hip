// FK kernel micro-benchmark (run on the GPU box): builds csrc/fk.hip's kernels into a standalone binary.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DFK_FAST_TRIG] -Idposer_amd/csrc tools/tune_fk.hip \
//         dposer_amd/csrc/fk.hip dposer_amd/csrc/elementwise.hip dposer_amd/csrc/gemm_launch.hip -o gpurun_out/tune_fk
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../include/dposer_hip.h"
int dposer_set_error(int code, const std::string& m) { fprintf(stderr, "%s\n", m.c_str()); return code; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int64_t B = argc > 1 ? atoll(argv[1]) : (1 << 20);
    const int n_out = argc > 2 ? atoi(argv[2]) : 22;
    static const int32_t parents[55] = {-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 15, 15, 15, 20, 25, 26,
                                        20, 28, 29, 20, 31, 32, 20, 34, 35, 20, 37, 38, 21, 40, 41, 21, 43, 44, 21, 46, 47, 21, 49, 50, 21, 52, 53};
    dposer_body_desc d = {55, 10475, 20, 21, 51};
    dposer_body_t h;
    if (dposer_body_create(&d, parents, &h)) return 1;
    float *pose, *jrest, *joints;
    CK(hipMalloc(&pose, B * 63 * 4)); CK(hipMalloc(&jrest, 55 * 3 * 4)); CK(hipMalloc(&joints, B * n_out * 3 * 4));
    std::vector<float> hp(B * 63), hj(55 * 3);
    srand(1);
    for (auto& v : hp) v = (rand() / (float)RAND_MAX - 0.5f) * 1.2f;
    for (auto& v : hj) v = (rand() / (float)RAND_MAX - 0.5f);
    CK(hipMemcpy(pose, hp.data(), hp.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(jrest, hj.data(), hj.size() * 4, hipMemcpyHostToDevice));
    const float* segs[7] = {nullptr, pose, nullptr, nullptr, nullptr, nullptr, nullptr};
    const int32_t segj[7] = {1, 21, 1, 1, 1, 15, 15};
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) dposer_fk_joints(h, segs, segj, 7, jrest, 0, nullptr, joints, nullptr, n_out, B, 0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < 20; ++i) dposer_fk_joints(h, segs, segj, 7, jrest, 0, nullptr, joints, nullptr, n_out, B, 0);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    const double us = ms * 1e3 / 20;
    std::vector<float> out(64 * n_out * 3);
    CK(hipMemcpy(out.data(), joints, out.size() * 4, hipMemcpyDeviceToHost));
    double cs = 0;
    for (float v : out) cs += v;
    printf("FK joints B=%lld n_out=%d: %.1f us  %.3g poses/s  %.1f GB/s (algorithmic %d B/pose)  checksum %.6f\n", (long long)B, n_out, us,
           B / (us * 1e-6), (252.0 + n_out * 12) * B / (us * 1e-6) / 1e9, 252 + n_out * 12, cs);
    return 0;
}
