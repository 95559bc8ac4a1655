// Probe: what bounds a GEMM tile's C stores — the chip's write bandwidth or one CU's store path?
// Every block (512 threads, one per CU) writes `tiles` 256 x 256 fp32 tiles (rows of 1 KiB, row stride `ld` floats) with
// nontemporal 16-byte stores, the way the store epilogue of gemm_nt_bf16.hip does; the grid size is the number of CUs that
// write at once.   hipcc --offload-arch=gfx950 -O3 scripts/probes/store_probe.hip -o gpurun_out/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) f32x4 gvec;

__global__ __launch_bounds__(512) void store_tiles(float* C, long ld, int tiles, int tiles_per_row, int nt) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const f32x4 v = {1.0f, 2.0f, 3.0f, (float)threadIdx.x};
    for (int t = 0; t < tiles; ++t) {
        const long tile = (long)t * gridDim.x + blockIdx.x;
        const long r0 = tile / tiles_per_row * 256, c0 = tile % tiles_per_row * 256;
        // wave w: rows w * 32 .. + 31 of the tile; per instruction 4 rows x 256 B?  no: as the epilogue — 2 rows x 512 B
        for (int i = 0; i < 16; ++i) {
            const long row = r0 + w * 32 + i * 2 + (lane >> 5);
            for (int h = 0; h < 2; ++h) {
                gvec* dst = (gvec*)(C + row * ld + c0 + h * 128 + (lane & 31) * 4);
                if (nt) __builtin_nontemporal_store(v, dst); else *dst = v;
            }
        }
    }
}

int main(int argc, char** argv) {
    const long ld = argc > 1 ? atol(argv[1]) : 32768;
    const int rows = 4096 * 12;
    float* C;
    if (hipMalloc(&C, (size_t)rows * ld * 4) != hipSuccess) return 1;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int tiles_per_row = 32768 / 256;
    for (int nt = 1; nt >= 0; --nt)
        for (int grid : {256, 128, 64, 32, 16, 8}) {
            const int tiles = 96;      // per block
            store_tiles<<<grid, 512>>>(C, ld, tiles, tiles_per_row, nt);
            hipEventRecord(e0);
            for (int r = 0; r < 3; ++r) store_tiles<<<grid, 512>>>(C, ld, tiles, tiles_per_row, nt);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
            const double bytes = (double)grid * tiles * 256 * 1024;
            printf("ld %ld nt %d blocks %3d: %7.1f GB/s total, %6.1f GB/s per block, %5.2f us per tile\n", ld, nt, grid, bytes / ms * 1e-6,
                   bytes / ms * 1e-6 / grid, ms * 1e3 / tiles);
        }
    return 0;
}
