// Semantics probe of ds_read_b64_tr_b16 (gfx950): which LDS elements does lane l receive, as a function of the
// addresses the 16 lanes of its group supply?  LDS holds lds16[i] = i.  Three address patterns:
//   0: lane l -> byte 8 l                      (contiguous chunks)
//   1: lane l -> byte 8 (l ^ 1)                (neighbouring lanes swapped)
//   2: lane l -> byte 128 (l >> 2 & 3) + 8 (l & 3) + 512 (l >> 4)   (four "pixel rows" of 128 B per 16-lane group)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void probe(int *out, int pat)
{
    __shared__ __attribute__((aligned(16))) short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x;
    int byte = pat == 0 ? 8 * l : pat == 1 ? 8 * (l ^ 1) : 128 * ((l >> 2) & 3) + 8 * (l & 3) + 512 * (l >> 4);
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3))) *)((__attribute__((address_space(3))) char *)lds + byte));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main()
{
    int *d, h[256];
    hipMalloc(&d, sizeof(h));
    for (int pat = 0; pat < 3; ++pat) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, pat);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("pattern %d\n", pat);
        for (int l = 0; l < 64; ++l) printf("  lane %2d: %4d %4d %4d %4d%s", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3], (l & 3) == 3 ? "\n" : "");
    }
    return 0;
}
