// ad-hoc probe: what bounds the random-column SpMV?  Not part of the product.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); exit(1);} } while(0)

__device__ __forceinline__ uint64_t mix64(uint64_t z){ z+=0x9E3779B97F4A7C15ULL; z=(z^(z>>30))*0xBF58476D1CE4E5B9ULL; z=(z^(z>>27))*0x94D049BB133111EBULL; return z^(z>>31);}

__global__ void gen(int64_t n, int per, int* ci, double* val){
  int64_t i = (int64_t)blockIdx.x*blockDim.x+threadIdx.x; if(i>=n*per) return;
  uint64_t h = mix64(i*7+1); ci[i] = (int)__umul64hi(h,(uint64_t)n); val[i] = (double)((h&3)+1);
}
__global__ void genrp(int64_t n, int per, int* rp){ int64_t i=(int64_t)blockIdx.x*blockDim.x+threadIdx.x; if(i<=n) rp[i]=(int)(i*per);}

template<int L, int MODE, int NT>
__global__ __launch_bounds__(256) void spmv(int n, const int* rp, const int* ci, const double* val, const double* x, double* y, int rows_per_block, int mask){
  constexpr int RPB=256/L; const int lane=threadIdx.x&(L-1), group=threadIdx.x/L;
  const int nb=gridDim.x,b=blockIdx.x; const int cid=((nb&7)==0)?(b&7)*(nb>>3)+(b>>3):b;
  long long r0=(long long)cid*rows_per_block; int rb=(int)(r0<n?r0:n); int re=(int)(r0+rows_per_block<n?r0+rows_per_block:n);
  for(int row=rb+group; row<re; row+=RPB){
    int s=rp[row], e=rp[row+1]; double sum=0;
    for(int k=s+lane;k<e;k+=L){
      double v = NT? __builtin_nontemporal_load(val+k): val[k];
      int c = NT? __builtin_nontemporal_load(ci+k): ci[k];
      if(MODE==0) sum += v*x[c];
      else if(MODE==1) sum += v*x[c & mask];       // x confined to an L2-sized window
      else sum += v*(double)c;                       // no gather at all: pure stream
    }
    for(int o=L/2;o>0;o>>=1) sum+=__shfl_xor(sum,o,64);
    if(lane==0) y[row]=sum;
  }
}

template<int L,int MODE,int NT> float run(int n,const int*rp,const int*ci,const double*val,const double*x,double*y,int grid_max,int mask,int reps){
  int rpb=256/L; long long groups=((long long)n+rpb-1)/rpb; int grid=(int)(groups<grid_max?groups:grid_max);
  long long per=((long long)n+grid-1)/grid; per=(per+rpb-1)/rpb*rpb; grid=(int)(((long long)n+per-1)/per);
  hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  spmv<L,MODE,NT><<<grid,256>>>(n,rp,ci,val,x,y,(int)per,mask); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for(int i=0;i<reps;i++) spmv<L,MODE,NT><<<grid,256>>>(n,rp,ci,val,x,y,(int)per,mask); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms,a,b)); return ms/reps;
}

int main(int argc,char**argv){
  int64_t n = argc>1? atoll(argv[1]) : 10000000; int per=50;
  int *rp,*ci; double *val,*x,*y;
  CK(hipMalloc(&rp,(n+1)*4)); CK(hipMalloc(&ci,n*per*4)); CK(hipMalloc(&val,n*per*8)); CK(hipMalloc(&x,n*8)); CK(hipMalloc(&y,n*8));
  gen<<<(unsigned)((n*per+255)/256),256>>>(n,per,ci,val); genrp<<<(unsigned)((n+256)/256),256>>>(n,per,rp); CK(hipMemset(x,0,n*8)); CK(hipDeviceSynchronize());
  double bytes = 12.0*n*per+20.0*n;
  #define R(L,MODE,NT,GM,MASK,name) { float ms=run<L,MODE,NT>((int)n,rp,ci,val,x,y,GM,MASK,5); printf("%-44s %8.3f ms  %8.1f GB/s\n",name,ms,bytes/ms*1e-6); fflush(stdout);} 
  R(32,0,1,2048,0,"gather L=32 nt grid2048 (current)");
  R(32,0,0,2048,0,"gather L=32 plain loads");
  R(16,0,1,2048,0,"gather L=16 nt");
  R(64,0,1,2048,0,"gather L=64 nt");
  R(8,0,1,2048,0,"gather L=8 nt");
  R(32,0,1,8192,0,"gather L=32 nt grid8192");
  R(32,0,1,65536,0,"gather L=32 nt grid65536");
  R(32,1,1,2048,(1<<18)-1,"x window 2 MB (L2-resident)");
  R(32,1,1,2048,(1<<17)-1,"x window 1 MB");
  R(32,1,1,2048,(1<<20)-1,"x window 8 MB (beyond one L2)");
  R(32,1,1,2048,(1<<22)-1,"x window 32 MB");
  R(32,2,1,2048,0,"no gather (stream only) L=32");
  R(16,2,1,2048,0,"no gather (stream only) L=16");
  R(64,2,1,2048,0,"no gather (stream only) L=64");
  R(32,2,0,2048,0,"no gather plain loads L=32");
  return 0;
}
