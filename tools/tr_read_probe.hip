// What does ds_read_b64_tr_b16 return?  (run on the GPU box)
// LDS holds u16 values = their own element index; every lane passes the byte address 8 * lane; the four 16-bit results per
// lane are printed.  Second pattern: addresses laid out as a [4 rows][16 cols] block per 16-lane group with a 64-byte row stride.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void probe(uint16_t* out, int mode) {
    __shared__ __attribute__((aligned(16))) volatile uint16_t lds[4096];   // volatile: the only reader is inline asm
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x;
    unsigned addr;
    if (mode == 0) addr = 8 * l;                                                   // 4 consecutive u16 per lane, lanes contiguous
    else { const int g = l >> 4, i = l & 15; addr = g * 1024 + (i >> 2) * 64 + (i & 3) * 8; }   // row = i>>2 (64-B stride), 4 cols per lane
    // `lds` is the only LDS object of the kernel => it starts at LDS offset 0 and `addr` is already the LDS byte address
    uint64_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    for (int k = 0; k < 4; ++k) out[l * 4 + k] = (uint16_t)(v >> (16 * k));
}

int main() {
    uint16_t* d;
    if (hipMalloc(&d, 64 * 4 * 2) != hipSuccess) return 1;
    for (int mode = 0; mode < 2; ++mode) {
        probe<<<1, 64>>>(d, mode);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
        uint16_t h[256];
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4u %4u %4u %4u%s", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], (l & 3) == 3 ? "\n" : "   ");
    }
    return 0;
}
