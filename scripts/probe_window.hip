// ad-hoc probe: rate of random 8-byte sc1 (agent-scope) gathers as a function of the window they fall into
// (80 MB = the whole C4 vector ... 2 MB): how much would level-major (window-local) polling buy the near part
// of the triangular solves?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); exit(1);} } while(0)
__device__ __forceinline__ uint64_t mix64(uint64_t z){ z+=0x9E3779B97F4A7C15ULL; z=(z^(z>>30))*0xBF58476D1CE4E5B9ULL; z=(z^(z>>27))*0x94D049BB133111EBULL; return z^(z>>31);}
// entry i gathers from a window of `win` doubles that slides with i (like rows of one group of levels)
__global__ void gen(int64_t n, int64_t win, int64_t cnt, int* ci){ int64_t i=(int64_t)blockIdx.x*blockDim.x+threadIdx.x; if(i<cnt){ int64_t base=(int64_t)((double)i/cnt*(n-win)); ci[i]=(int)(base+(int64_t)__umul64hi(mix64(i*7+1),(uint64_t)win)); } }
template<int MODE> __global__ __launch_bounds__(256) void gather(int64_t cnt, const int* ci, const double* x, double* out){
  int64_t i=(int64_t)blockIdx.x*blockDim.x+threadIdx.x, st=(int64_t)gridDim.x*blockDim.x; double acc=0;
  for(;i<cnt;i+=st){ int c=__builtin_nontemporal_load(ci+i); double v;
    if(MODE==0) v=x[c]; else v=__hip_atomic_load(x+c,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_AGENT);
    acc+=v; }
  if(acc==1.2345) out[0]=acc;
}
int main(){
  int64_t n=10000000, cnt=100000000; int* ci; double *x,*out;
  CK(hipMalloc(&ci,cnt*4)); CK(hipMalloc(&x,n*8)); CK(hipMalloc(&out,8)); CK(hipMemset(x,0,n*8));
  for (int64_t win : {10000000LL, 2000000LL, 1000000LL, 500000LL, 250000LL, 100000LL}) {
    gen<<<(unsigned)((cnt+255)/256),256>>>(n,win,cnt,ci); CK(hipDeviceSynchronize());
    for (int mode=0; mode<2; mode++) {
      hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
      if(mode) gather<1><<<4096,256>>>(cnt,ci,x,out); else gather<0><<<4096,256>>>(cnt,ci,x,out);
      CK(hipDeviceSynchronize()); CK(hipEventRecord(a));
      for(int r=0;r<3;r++){ if(mode) gather<1><<<4096,256>>>(cnt,ci,x,out); else gather<0><<<4096,256>>>(cnt,ci,x,out); }
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms,a,b)); ms/=3;
      printf("window %5.1f MB  %-10s %8.3f ms  %7.1f G gathers/s\n", win*8/1e6, mode?"sc1 load":"plain load", ms, cnt/ms*1e-6); fflush(stdout);
    }
  }
  return 0;
}
