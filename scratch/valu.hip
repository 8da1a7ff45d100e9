#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));
#define N_IT 4096
template<int MODE> __global__ __launch_bounds__(256) void k(float* out, float a, float b){
  float x0=threadIdx.x*1e-3f, x1=x0+1, x2=x0+2, x3=x0+3, x4=x0+4,x5=x0+5,x6=x0+6,x7=x0+7;
  float2v p0={x0,x1},p1={x2,x3},p2={x4,x5},p3={x6,x7}; float2v A={a,a},B={b,b};
  for(int i=0;i<N_IT;i++){
    if(MODE==0){ // 8 scalar fma
      x0=__builtin_fmaf(x0,a,b);x1=__builtin_fmaf(x1,a,b);x2=__builtin_fmaf(x2,a,b);x3=__builtin_fmaf(x3,a,b);
      x4=__builtin_fmaf(x4,a,b);x5=__builtin_fmaf(x5,a,b);x6=__builtin_fmaf(x6,a,b);x7=__builtin_fmaf(x7,a,b);
    } else if(MODE==1){ // 4 packed fma = 8 fma
      p0=__builtin_elementwise_fma(p0,A,B);p1=__builtin_elementwise_fma(p1,A,B);p2=__builtin_elementwise_fma(p2,A,B);p3=__builtin_elementwise_fma(p3,A,B);
    } else if(MODE==2){ // 8 scalar mul
      x0*=a;x1*=a;x2*=a;x3*=a;x4*=a;x5*=a;x6*=a;x7*=a;
    } else if (MODE==3){ // 8 rcp
      x0=__builtin_amdgcn_rcpf(x0);x1=__builtin_amdgcn_rcpf(x1);x2=__builtin_amdgcn_rcpf(x2);x3=__builtin_amdgcn_rcpf(x3);
      x4=__builtin_amdgcn_rcpf(x4);x5=__builtin_amdgcn_rcpf(x5);x6=__builtin_amdgcn_rcpf(x6);x7=__builtin_amdgcn_rcpf(x7);
    } else if (MODE==4){ // 8 IEEE div
      x0=a/x0;x1=a/x1;x2=a/x2;x3=a/x3;x4=a/x4;x5=a/x5;x6=a/x6;x7=a/x7;
    } else if (MODE==5){ // 8 IEEE sqrt
      x0=sqrtf(x0+a);x1=sqrtf(x1+a);x2=sqrtf(x2+a);x3=sqrtf(x3+a);x4=sqrtf(x4+a);x5=sqrtf(x5+a);x6=sqrtf(x6+a);x7=sqrtf(x7+a);
    }
  }
  out[blockIdx.x*256+threadIdx.x]=x0+x1+x2+x3+x4+x5+x6+x7+p0.x+p0.y+p1.x+p1.y+p2.x+p2.y+p3.x+p3.y;
}
template<int MODE> void run(const char* name, int blocksPerCU){
  float* d; hipMalloc(&d, 256*256*16*4*4); hipEvent_t e0,e1; hipEventCreate(&e0);hipEventCreate(&e1);
  int grid=256*blocksPerCU;
  k<MODE><<<grid,256>>>(d,1.0001f,0.5f); hipDeviceSynchronize();
  hipEventRecord(e0); k<MODE><<<grid,256>>>(d,1.0001f,0.5f); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms,e0,e1);
  double waves=grid*4.0; double ops=waves*N_IT*8.0; // 8 float-ops (per lane) per iter
  double simdcycles = ms*1e-3*2.4e9*1024;
  printf("%-14s blocks/CU %d: %.3f ms  -> %.2f SIMD-cycles (at 2.4GHz) per wave-level scalar-op-equivalent\n", name, blocksPerCU, ms, simdcycles/ops);
}
int main(){ for(int b: {1,2,4}){ run<0>("fma",b); run<1>("pk_fma",b); run<2>("mul",b); run<3>("rcp",b); run<4>("ieee div",b); run<5>("ieee sqrt",b);} }
