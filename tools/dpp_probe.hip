#include <hip/hip_runtime.h>
typedef unsigned u32; typedef unsigned long long u64;
__device__ __forceinline__ void swap32(u32 &a, u32 &b) { auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false); a = r[0]; b = r[1]; }
__device__ __forceinline__ void swap16(u32 &a, u32 &b) { auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false); a = r[0]; b = r[1]; }
__global__ void k(u32 *o) {
  u32 a = threadIdx.x, b = 1000 + threadIdx.x;
  swap32(a, b);
  u32 c = threadIdx.x, d = 1000 + threadIdx.x;
  swap16(c, d);
  u32 e = __builtin_amdgcn_update_dpp(threadIdx.x, 2000 + threadIdx.x, 0x128, 0xF, 0xC, false);   // row_ror:8, banks 2,3
  u32 f = __builtin_amdgcn_update_dpp(threadIdx.x, 3000 + threadIdx.x, 0x114, 0xF, 0xA, false);   // row_shr:4, banks 1,3
  u32 g = __builtin_amdgcn_update_dpp(threadIdx.x, 4000 + threadIdx.x, 0x104, 0xF, 0x5, false);   // row_shl:4, banks 0,2
  u32 h = __builtin_amdgcn_mov_dpp(5000 + threadIdx.x, 0x4E, 0xF, 0xF, true);                      // quad_perm [2,3,0,1]
  u32 i = __builtin_amdgcn_mov_dpp(6000 + threadIdx.x, 0xB1, 0xF, 0xF, true);                      // quad_perm [1,0,3,2]
  u32 *p = o + threadIdx.x * 9; p[0]=a; p[1]=b; p[2]=c; p[3]=d; p[4]=e; p[5]=f; p[6]=g; p[7]=h; p[8]=i;
}
int main() { u32 *d; hipMalloc(&d, 64*9*4); k<<<1,64>>>(d); u32 h[64*9]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  for (int t = 0; t < 64; t++) { printf("%2d:", t); for (int j = 0; j < 9; j++) printf(" %4u", h[t*9+j]); printf("\n"); } return 0; }
