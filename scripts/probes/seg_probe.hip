// Probe: does the segment length of a transposing pass's writes bound it?  The traffic of jumprelu_fwd_bf16_kernel — a [rows][n]
// 16-bit tensor read once, written once row-major and once transposed ([n][rows]) — without the arithmetic and without the LDS
// transposition (register values go to the transposed addresses): per block a TR x TC tile, row-major segments of 2 TC bytes,
// transposed segments of 2 TR bytes.   hipcc --offload-arch=gfx950 -O3 scripts/probes/seg_probe.hip -o gpurun_out/seg_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int TR, int TC, bool ROWMAJOR, bool TRANSPOSED>
__global__ __launch_bounds__(256) void pass(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, uint16_t* __restrict__ yT, int rows, long n) {
    const long c0 = (long)blockIdx.x * TC;
    const int r0 = blockIdx.y * TR;
    constexpr int TPR = TC / 4, RPP = 256 / TPR, NP = TR / RPP;          // threads per row, rows per pass, passes
    const int tc = threadIdx.x % TPR, tr = threadIdx.x / TPR;
    u32x2 v[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) v[p] = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(x + (long)(r0 + p * RPP + tr) * n + c0 + 4 * tc));
    uint32_t acc = 0;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        acc ^= v[p][0] + v[p][1];
        if (ROWMAJOR) __builtin_nontemporal_store(v[p], reinterpret_cast<u32x2*>(y + (long)(r0 + p * RPP + tr) * n + c0 + 4 * tc));
    }
    if (TRANSPOSED) {
        constexpr int TPC = TR / 8, CPP = 256 / TPC, NQ = TC / CPP;      // threads per column, columns per pass, passes
        const int qr = threadIdx.x % TPC, qc = threadIdx.x / TPC;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const u32x4 o = {acc, acc + 1, acc + 2, acc + (uint32_t)q};
            __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(yT + (c0 + q * CPP + qc) * rows + r0 + 8 * qr));
        }
    }
}

template <int TR, int TC, bool RM, bool TP>
void run(const uint16_t* x, uint16_t* y, uint16_t* yT, int rows, long n, const char* what) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid((unsigned)(n / TC), (unsigned)(rows / TR));
    pass<TR, TC, RM, TP><<<grid, 256>>>(x, y, yT, rows, n);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) pass<TR, TC, RM, TP><<<grid, 256>>>(x, y, yT, rows, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double bytes = (double)rows * n * 2 * (1 + (RM ? 1 : 0) + (TP ? 1 : 0));
    printf("%-22s tile %3d x %3d (row segments %4d B, transposed %4d B): %7.1f us  %6.0f GB/s\n", what, TR, TC, 2 * TC, 2 * TR, ms * 1e3, bytes / ms * 1e-6);
}

int main() {
    const int rows = 4096; const long n = 12L * 32768;
    uint16_t *x, *y, *yT;
    const size_t sz = (size_t)rows * n * 2;
    if (hipMalloc(&x, sz) != hipSuccess || hipMalloc(&y, sz) != hipSuccess || hipMalloc(&yT, sz) != hipSuccess) return 1;
    hipMemset(x, 1, sz);
    for (int rep = 0; rep < 2; ++rep) {
        run<64, 256, true, false>(x, y, yT, rows, n, "read + row-major");
        run<64, 256, false, true>(x, y, yT, rows, n, "read + transposed");
        run<128, 128, false, true>(x, y, yT, rows, n, "read + transposed");
        run<256, 64, false, true>(x, y, yT, rows, n, "read + transposed");
        run<64, 256, true, true>(x, y, yT, rows, n, "read + both");
        run<128, 128, true, true>(x, y, yT, rows, n, "read + both");
        run<128, 256, true, true>(x, y, yT, rows, n, "read + both");
        run<256, 128, true, true>(x, y, yT, rows, n, "read + both");
        run<256, 64, true, true>(x, y, yT, rows, n, "read + both");
        run<32, 256, true, true>(x, y, yT, rows, n, "read + both");
    }
    return 0;
}
