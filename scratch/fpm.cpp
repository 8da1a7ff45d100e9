#include "szg/fpmath.h"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
static double ulp_err(float got, double want){ if(std::isnan(got)&&std::isnan(want)) return 0; if (std::isinf(want) || want==0) return got==(float)want?0:1e9;
  float w=(float)want; int e; frexp(want,&e); double ulp=ldexp(1.0, e-24); if (fabs(want) < 1.17549435e-38) ulp = ldexp(1.0,-149); return fabs((double)got-want)/ulp; }
int main(){ std::mt19937_64 g(3);
 auto sweep=[&](const char* n, float lo, float hi, auto f, auto ref){ std::uniform_real_distribution<double> d(lo,hi); double mx=0; float arg=0; for(int i=0;i<4000000;i++){ float x=(float)d(g); double e=ulp_err(f(x),ref((double)x)); if(e>mx){mx=e;arg=x;} } printf("%-10s [%g,%g] max err %.3f ulp at %.9g\n",n,lo,hi,mx,arg); };
 sweep("exp",-87,88,[](float x){return szg_expf(x);},[](double x){return exp(x);});
 sweep("exp",-1,1,[](float x){return szg_expf(x);},[](double x){return exp(x);});
 sweep("exp",-104,-87,[](float x){return szg_expf(x);},[](double x){return exp(x);});
 sweep("log",1e-3,10,[](float x){return szg_logf(x);},[](double x){return log(x);});
 sweep("log",0.5,2,[](float x){return szg_logf(x);},[](double x){return log(x);});
 sweep("log",0,1e-37,[](float x){return szg_logf(x);},[](double x){return log(x);});
 sweep("log",1e3,3e38,[](float x){return szg_logf(x);},[](double x){return log(x);});
 sweep("sin",-10,10,[](float x){return szg_sinf(x);},[](double x){return sin(x);});
 sweep("cos",-10,10,[](float x){return szg_cosf(x);},[](double x){return cos(x);});
 sweep("sin",-0.01,0.01,[](float x){return szg_sinf(x);},[](double x){return sin(x);});
 sweep("sin",-1000,1000,[](float x){return szg_sinf(x);},[](double x){return sin(x);});
 sweep("asin",-1,1,[](float x){return szg_asinf(x);},[](double x){return asin(x);});
 sweep("acos",-1,1,[](float x){return szg_acosf(x);},[](double x){return acos(x);});
 sweep("acos",0.99,1,[](float x){return szg_acosf(x);},[](double x){return acos(x);});
 sweep("pow1.2",0,20,[](float x){return szg_powf(x,1.2f);},[](double x){return pow(x,1.2);});
 sweep("pow5",0,1,[](float x){return szg_powf(x,5.f);},[](double x){return pow(x,5.0);});
 sweep("pow1.5",0.3,3.3,[](float x){return szg_powf(x,1.5f);},[](double x){return pow(x,1.5);});
 sweep("pow160",0.7,1,[](float x){return szg_powf(x,160.f);},[](double x){return pow(x,160.0);});
 printf("exp(0)=%g exp(-inf)=%g exp(inf)=%g log(1)=%g pow(0,5)=%g pow(1,160)=%g sin(0)=%g cos(0)=%g asin(1)=%.9g acos(-1)=%.9g acos(1)=%g\n", szg_expf(0),szg_expf(-INFINITY),szg_expf(INFINITY),szg_logf(1),szg_powf(0,5),szg_powf(1,160),szg_sinf(0),szg_cosf(0),szg_asinf(1),szg_acosf(-1),szg_acosf(1));
}
