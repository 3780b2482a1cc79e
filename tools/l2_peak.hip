// l2_peak.hip — what the L2 -> CU path of MI355X delivers for a weight-like stream: every workgroup (one per CU, 256 threads = 4
// waves, 16 bytes per lane per load) re-reads the SAME `bytes`-sized buffer `reps` times (after the first pass it lives in each XCD's
// L2: 4 MB per XCD), the way every CU of the Winograd kernels streams the layer's weight fragments.  Prints TB/s over all CUs.
//   hipcc --offload-arch=gfx950 -O3 tools/l2_peak.hip -o /tmp/l2_peak && /tmp/l2_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int INFLIGHT>
__global__ __launch_bounds__(256) void stream_kernel(const u32x4* __restrict__ buf, long n16, int reps, unsigned* out, int stagger) {
    u32x4 acc = {0u, 0u, 0u, 0u};
    const long start = stagger ? ((long)blockIdx.x * 9973 * 64) % n16 : 0;      // stagger: CUs walk the buffer out of phase
    for (int r = 0; r < reps; ++r) {
        for (long i = threadIdx.x; i < n16; i += 256 * INFLIGHT) {
            u32x4 v[INFLIGHT];
#pragma unroll
            for (int k = 0; k < INFLIGHT; ++k) {
                long j = i + (long)k * 256 + start;
                if (j >= n16) j -= n16;
                v[k] = buf[j < n16 ? j : 0];
            }
#pragma unroll
            for (int k = 0; k < INFLIGHT; ++k) acc ^= v[k];
        }
    }
    if (acc[0] == 0x12345678u) out[0] = acc[1] ^ acc[2] ^ acc[3];
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    unsigned* out;
    hipMalloc(&out, 4);
    for (long bytes : {1l << 20, 2l << 20, 9l << 20, 64l << 20}) {
        u32x4* buf;
        hipMalloc(&buf, bytes);
        hipMemset(buf, 1, bytes);
        const long n16 = bytes / 16;
        for (int stagger = 0; stagger < 2; ++stagger) {
            const int reps = (int)((256l << 20) / bytes) + 1;
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            for (int w = 0; w < 2; ++w) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(stream_kernel<8>, dim3(cus), dim3(256), 0, 0, buf, n16, reps, out, stagger);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double tb = (double)bytes * reps * cus / (ms * 1e-3) / 1e12;
            printf("buffer %3ld MB  stagger %d  %d CUs x 256 threads, 8 x 16 B in flight per lane: %.2f TB/s  (%.1f B/clk/CU at 2.1 GHz)\n", bytes >> 20, stagger,
                   cus, tb, tb * 1e12 / cus / 2.1e9);
        }
        hipFree(buf);
    }
    return 0;
}
