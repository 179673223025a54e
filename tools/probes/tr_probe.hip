// Hardware probe: what does ds_read_b64_tr_b16 return?  LDS holds u16 value == element index;
// lane l supplies byte address 8*l (elements 4l..4l+3).  Prints, per lane, the 4 element indices read.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__global__ void probe(unsigned short* out, int stride_bytes) {
    __shared__ unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    unsigned addr = (unsigned)(size_t)lds + threadIdx.x * stride_bytes;
    u32x2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    out[threadIdx.x * 4 + 0] = r[0] & 0xffff; out[threadIdx.x * 4 + 1] = r[0] >> 16;
    out[threadIdx.x * 4 + 2] = r[1] & 0xffff; out[threadIdx.x * 4 + 3] = r[1] >> 16;
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    unsigned short h[256];
    for (int stride : {8, 32}) {
        probe<<<1, 64>>>(d, stride);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("stride_bytes=%d\n", stride);
        for (int l = 0; l < 64; l++) printf("lane %2d: %4d %4d %4d %4d%s", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3], (l % 4 == 3) ? "\n" : "   |  ");
    }
    return 0;
}
