#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
#include <cstring>
__global__ void k(const float* a, const float* b, float* o, int n){
  int i = blockIdx.x*blockDim.x+threadIdx.x; if(i>=n) return;
  o[i] = a[i]/b[i]; o[n+i] = sqrtf(fabsf(a[i])); o[2*n+i] = 1.0f/sqrtf(fabsf(b[i])); o[3*n+i]=expf(-fabsf(a[i])); o[4*n+i]=sinf(a[i]); o[5*n+i]=cosf(a[i]);
  o[6*n+i]=asinf(fminf(fabsf(a[i])*0.1f,1.f)); o[7*n+i]=acosf(fminf(fabsf(a[i])*0.1f,1.f)); o[8*n+i]=powf(fabsf(a[i])*0.1f, 1.2f); o[9*n+i] = a[i]*b[i]+a[i];
  o[10*n+i] = fmaf(a[i],b[i],a[i]); o[11*n+i]=floorf(a[i]*3.7f);
}
int main(){ int n=1<<20; std::vector<float> a(n),b(n),o(12*n); std::mt19937 g(1); std::uniform_real_distribution<float> d(-10,10);
 for(int i=0;i<n;i++){a[i]=d(g); b[i]=d(g);} float *da,*db,*dout; hipMalloc(&da,n*4);hipMalloc(&db,n*4);hipMalloc(&dout,12*n*4);
 hipMemcpy(da,a.data(),n*4,hipMemcpyHostToDevice);hipMemcpy(db,b.data(),n*4,hipMemcpyHostToDevice);
 k<<<n/256,256>>>(da,db,dout,n); hipMemcpy(o.data(),dout,12*n*4,hipMemcpyDeviceToHost);
 const char* names[]={"div","sqrt","1/sqrt","exp","sin","cos","asin","acos","pow1.2","mul+add","fma","floor"};
 for(int f=0;f<12;f++){ long diff=0; int maxulp=0; for(int i=0;i<n;i++){ float x=a[i],y=b[i],r;
   volatile float t;
   switch(f){case 0:r=x/y;break;case 1:r=sqrtf(fabsf(x));break;case 2:r=1.0f/sqrtf(fabsf(y));break;case 3:r=expf(-fabsf(x));break;case 4:r=sinf(x);break;case 5:r=cosf(x);break;
   case 6:r=asinf(fminf(fabsf(x)*0.1f,1.f));break;case 7:r=acosf(fminf(fabsf(x)*0.1f,1.f));break;case 8:r=powf(fabsf(x)*0.1f,1.2f);break;case 9: t=x*y; r=t+x;break;case 10:r=fmaf(x,y,x);break;default:r=floorf(x*3.7f);}
   float q=o[f*n+i]; if(memcmp(&q,&r,4)){diff++; int qi,ri; memcpy(&qi,&q,4);memcpy(&ri,&r,4); int u=abs(qi-ri); if(u>maxulp)maxulp=u;} }
   printf("%-8s mismatches %ld / %d  max ulp %d\n",names[f],diff,n,maxulp);} }
