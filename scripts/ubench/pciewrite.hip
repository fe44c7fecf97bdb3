// How fast do a kernel's stores reach page-locked host memory, by the size of the contiguous pieces they come in?
// (a kept canvas, mi355_opts::keep_canvas: the tile kernel's blocks write 16 rows of 64 bytes each, 7.6 KB apart)
//     hipcc -O2 --offload-arch=gfx950 -o /tmp/pciewrite scripts/ubench/pciewrite.hip && /tmp/pciewrite
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
// piece p of `bytes` bytes lies at dst + p * stride; a block of 256 threads writes 16 pieces of 64 B (a tile), or whole pieces
__global__ void k_write(uint4 *dst, size_t stride16, int piece16, size_t n_pieces, int pieces_per_block)
{
    const size_t p0 = (size_t)blockIdx.x * pieces_per_block;
    for (int i = threadIdx.x; i < pieces_per_block * piece16; i += blockDim.x) {
        const size_t p = p0 + i / piece16;
        // (piece p: row p % 8192 of the buffer, column p / 8192 -- neighbours in p are a row apart, like the rows of a tile)
        if (p < n_pieces) dst[(p % 8192u) * stride16 + (p / 8192u) * piece16 + i % piece16] = make_uint4(1u, 2u, 3u, (uint32_t)p);
    }
}
int main()
{
    const size_t total = 64u << 20;
    void *h = nullptr, *d = nullptr, *dev = nullptr;
    if (hipHostMalloc(&h, total, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) return 1;
    memset(h, 0, total);
    (void)hipHostGetDevicePointer(&d, h, 0);
    (void)hipMalloc(&dev, total);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int target = 0; target < 2; target++)
        for (int piece : {32, 64, 128, 256, 512, 1024}) {
            const int piece16 = piece / 16;
            const size_t stride16 = 7680 / 16;                   // a 1920-pixel row
            const size_t payload = 2u << 20;                      // what a kept chessboard frame sends
            const size_t n_pieces = payload / piece;
            const int ppb = piece <= 64 ? 16 : (piece <= 256 ? 4 : 1);
            const unsigned blocks = (unsigned)((n_pieces + ppb - 1) / ppb);
            uint4 *dst = (uint4 *)(target ? dev : d);
            for (int w = 0; w < 3; w++) hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, 0, dst, stride16, piece16, n_pieces, ppb);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0, 0);
            for (int w = 0; w < 20; w++) hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, 0, dst, stride16, piece16, n_pieces, ppb);
            (void)hipEventRecord(e1, 0);
            (void)hipDeviceSynchronize();
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            printf("%s, pieces of %4d B (%6zu of them, 2 MB): %7.1f us per pass, %6.1f GB/s\n", target ? "device memory" : "host memory  ", piece, n_pieces, ms / 20 * 1e3, payload / (ms / 20 * 1e-3) / 1e9);
        }
    return 0;
}
