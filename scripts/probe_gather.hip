// ad-hoc probe: can a random 8-byte gather be made to move less than a 128-byte line?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); exit(1);} } while(0)
__device__ __forceinline__ uint64_t mix64(uint64_t z){ z+=0x9E3779B97F4A7C15ULL; z=(z^(z>>30))*0xBF58476D1CE4E5B9ULL; z=(z^(z>>27))*0x94D049BB133111EBULL; return z^(z>>31);}
__global__ void gen(int64_t n, int64_t cnt, int* ci){ int64_t i=(int64_t)blockIdx.x*blockDim.x+threadIdx.x; if(i<cnt) ci[i]=(int)__umul64hi(mix64(i*7+1),(uint64_t)n);} 
template<int MODE> __global__ __launch_bounds__(256) void gather(int64_t cnt, const int* ci, const double* x, double* out){
  int64_t i=(int64_t)blockIdx.x*blockDim.x+threadIdx.x, st=(int64_t)gridDim.x*blockDim.x; double acc=0;
  for(;i<cnt;i+=st){ int c=__builtin_nontemporal_load(ci+i); double v;
    if(MODE==0) v=x[c];
    else if(MODE==1) v=__builtin_nontemporal_load(x+c);
    else if(MODE==2) v=__hip_atomic_load(x+c,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_AGENT);
    else v=__hip_atomic_load(x+c,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_SYSTEM);
    acc+=v; }
  if(acc==1.2345) out[0]=acc;
}
template<int MODE> void run(const char* name,int64_t cnt,const int*ci,const double*x,double*out){
  hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  gather<MODE><<<4096,256>>>(cnt,ci,x,out); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for(int r=0;r<3;r++) gather<MODE><<<4096,256>>>(cnt,ci,x,out); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms,a,b)); ms/=3; printf("%-46s %8.3f ms  %7.1f G gathers/s\n",name,ms,cnt/ms*1e-6); fflush(stdout);
}
int main(){
  int64_t n=10000000, cnt=250000000; int* ci; double *x,*xu,*xf,*out;
  CK(hipMalloc(&ci,cnt*4)); CK(hipMalloc(&x,n*8)); CK(hipMalloc(&out,8));
  CK(hipExtMallocWithFlags((void**)&xu,n*8,hipDeviceMallocUncached));
  CK(hipExtMallocWithFlags((void**)&xf,n*8,hipDeviceMallocFinegrained));
  gen<<<(unsigned)((cnt+255)/256),256>>>(n,cnt,ci); CK(hipMemset(x,0,n*8)); CK(hipMemset(xu,0,n*8)); CK(hipMemset(xf,0,n*8)); CK(hipDeviceSynchronize());
  run<0>("plain load, hipMalloc",cnt,ci,x,out);
  run<1>("nontemporal load, hipMalloc",cnt,ci,x,out);
  run<2>("agent-scope relaxed atomic load (sc1)",cnt,ci,x,out);
  run<3>("system-scope relaxed atomic load (sc0 sc1)",cnt,ci,x,out);
  run<0>("plain load, hipDeviceMallocUncached",cnt,ci,xu,out);
  run<1>("nontemporal load, Uncached",cnt,ci,xu,out);
  run<3>("system-scope load, Uncached",cnt,ci,xu,out);
  run<0>("plain load, hipDeviceMallocFinegrained",cnt,ci,xf,out);
  run<3>("system-scope load, Finegrained",cnt,ci,xf,out);
  return 0;
}
