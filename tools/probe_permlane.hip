#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* a, unsigned* b){
  unsigned x=1000+threadIdx.x, y=2000+threadIdx.x;
  auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
  a[threadIdx.x]=r[0]; b[threadIdx.x]=r[1];
}
int main(){ unsigned *a,*b; (void)hipMalloc(&a,256); (void)hipMalloc(&b,256); hipLaunchKernelGGL(k,1,64,0,0,a,b); unsigned ha[64],hb[64]; (void)hipMemcpy(ha,a,256,hipMemcpyDeviceToHost); (void)hipMemcpy(hb,b,256,hipMemcpyDeviceToHost);
 for(int i=0;i<64;i+=8) printf("lane %2d: r0=%u r1=%u\n", i, ha[i], hb[i]); return 0; }
