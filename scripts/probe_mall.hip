// ad-hoc probe: does a producer->consumer buffer that is REUSED stay in the 256 MiB Infinity Cache?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); exit(1);} } while(0)
// producer: reads 10 B per element (8 from a, 2 from b), writes 8 B to P
__global__ __launch_bounds__(1024) void prod(const double2* a, const ushort2* b, double2* P, long n2){
  long i=(long)blockIdx.x*blockDim.x+threadIdx.x, st=(long)gridDim.x*blockDim.x;
  for(;i<n2;i+=st){ double2 v=a[i]; ushort2 c=b[i]; double2 o; o.x=v.x*(double)c.x; o.y=v.y*(double)c.y; P[i]=o; }
}
// consumer: reads 8 B from P and 2 B from b
__global__ __launch_bounds__(1024) void cons(const double2* P, const ushort2* b, double* out, long n2){
  long i=(long)blockIdx.x*blockDim.x+threadIdx.x, st=(long)gridDim.x*blockDim.x; double acc=0;
  for(;i<n2;i+=st){ double2 v=P[i]; ushort2 c=b[i]; acc+=v.x*(double)c.x+v.y; }
  if(acc==1.2345) out[0]=acc;
}
int main(){
  const long TOT = 500000000L;            // elements in the big streams (4 GB of doubles, 1 GB of u16)
  double *a,*P,*out; unsigned short *b,*b2;
  CK(hipMalloc(&a,TOT*8)); CK(hipMalloc(&b,TOT*2)); CK(hipMalloc(&b2,TOT*2)); CK(hipMalloc(&P,TOT*8)); CK(hipMalloc(&out,8));
  CK(hipMemset(a,0,TOT*8)); CK(hipMemset(b,0,TOT*2)); CK(hipMemset(b2,0,TOT*2)); CK(hipDeviceSynchronize());
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  long sizes_mb[] = {16,32,64,96,128,192,256,512,4000};
  for(long smb : sizes_mb){
    long S = smb*1000000L/8;              // elements per slab
    if(S>TOT) S=TOT;
    long nslab = TOT / S;
    for(int reuse=0; reuse<2; reuse++){
      // warm
      for(int rep=0; rep<2; rep++){
        if(rep==1) CK(hipEventRecord(e0));
        for(long s=0;s<nslab;s++){
          double* Ps = reuse ? P : P + s*S;          // reuse=1: the same S elements every slab
          prod<<<512,1024>>>((const double2*)(a+s*S),(const ushort2*)(b+s*S),(double2*)Ps,S/2);
          cons<<<512,1024>>>((const double2*)Ps,(const ushort2*)(b2+s*S),out,S/2);
        }
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1));
      double bytes = (double)nslab*S*(10.0+8.0+10.0);
      printf("slab %5ld MB  %s  total %8.3f ms  (%6.1f GB/s over 28 B/elem; per-pair launches %ld)\n", smb, reuse?"P reused ":"P streamed", ms, bytes/ms*1e-6, nslab);
      fflush(stdout);
    }
  }
  return 0;
}
