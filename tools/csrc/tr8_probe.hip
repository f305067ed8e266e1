#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
// what does ds_read_b64_tr_b8 return?  LDS holds a [64 rows][16 cols] byte matrix, byte = (row << 4 | col) & 0xff for rows < 16 of each
// 16-row block (block id in the next 256 bytes...).  Lane t of a 16-lane group addresses row t/2, columns 8 (t%2) .. +7 of ITS group's block.
__global__ void k(uint32_t* out) {
    __shared__ __attribute__((aligned(16))) unsigned char m[4 * 256];
    const int lane = threadIdx.x;
    for (int i = lane; i < 4 * 256; i += 64) m[i] = (unsigned char)(i & 0xff);     // block g: byte = row*16 + col (row < 16)
    __syncthreads();
    const int g = lane >> 4, t = lane & 15;
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)m + g * 256 + (t >> 1) * 16 + (t & 1) * 8;
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
    out[lane * 2] = (uint32_t)v; out[lane * 2 + 1] = (uint32_t)(v >> 32);
}
int main() {
    uint32_t* d; hipMalloc(&d, 128 * 4);
    k<<<1, 64>>>(d);
    uint32_t h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int b = 0; b < 8; ++b) { unsigned v = (h[l * 2 + b / 4] >> (8 * (b % 4))) & 0xff; printf(" (r%d,c%d)", v >> 4, v & 15); }
        printf("\n");
    }
    return 0;
}
