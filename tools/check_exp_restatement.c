#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "exp_tab.h"
static inline uint64_t asuint64(double x){uint64_t u; memcpy(&u,&x,8); return u;}
static inline double asdouble(uint64_t u){double x; memcpy(&x,&u,8); return x;}
static inline uint32_t top12(double x){return asuint64(x)>>52;}
#define N 128
static const double InvLn2N = 0x1.71547652b82fep0 * N, NegLn2hiN = -0x1.62e42fefa0000p-8, NegLn2loN = -0x1.cf79abc9e3b3ap-47, Shift = 0x1.8p52;
static const double C2 = 0x1.ffffffffffdbdp-2, C3 = 0x1.555555555543cp-3, C4 = 0x1.55555cf172b91p-5, C5 = 0x1.1111167a4d017p-7;
#ifdef USE_FMA
#define FMA(a,b,c) fma(a,b,c)
#else
#define FMA(a,b,c) ((a)*(b)+(c))
#endif
#ifdef SC_NOFMA
#define SC(s,t) ((s)+(s)*(t))
#define SCLO(s,t,y) ((s)-(y)+(s)*(t))
#else
#define SC(s,t) FMA(s,t,s)
#define SCLO(s,t,y) FMA(s,t,(s)-(y))
#endif
static double specialcase(double tmp, uint64_t sbits, uint64_t ki){
  double scale,y;
  if ((ki & 0x80000000)==0){ sbits -= 1009ull<<52; scale=asdouble(sbits); y = 0x1p1009 * SC(scale,tmp); return y; }
  sbits += 1022ull<<52; scale=asdouble(sbits); y = SC(scale,tmp);
  if (y < 1.0){ double hi,lo; lo = SCLO(scale,tmp,y); hi = 1.0 + y; lo = 1.0 - hi + y + lo; y = (hi+lo) - 1.0; if (y==0.0) y=0.0; }
  y = 0x1p-1022 * y; return y; }
static double myexp(double x){
  uint32_t abstop = top12(x) & 0x7ff;
  if (abstop - top12(0x1p-54) >= top12(512.0) - top12(0x1p-54)) {
    if (abstop - top12(0x1p-54) >= 0x80000000) return 1.0 + x;
    if (abstop >= top12(1024.0)) { if (asuint64(x)==asuint64(-INFINITY)) return 0.0; if (abstop >= top12(INFINITY)) return 1.0+x; if (asuint64(x)>>63) return 0.0; else return INFINITY; }
    abstop = 0; }
  double kd = FMA(InvLn2N, x, Shift);
  uint64_t ki = asuint64(kd); kd -= Shift;
  double r = FMA(kd, NegLn2loN, FMA(kd, NegLn2hiN, x));
  uint64_t idx = 2*(ki % N), top = ki << (52-7);
  double tail = asdouble(EXP_TAB[idx]); uint64_t sbits = EXP_TAB[idx+1] + top;
  double r2 = r*r;
  double tmp = FMA(r2*r2, FMA(r,C5,C4), FMA(r2, FMA(r,C3,C2), tail + r));
  if (abstop==0) return specialcase(tmp,sbits,ki);
  double scale = asdouble(sbits); return FMA(scale,tmp,scale); }
int main(){ uint64_t s=88172645463325252ull; long bad=0,n=0;
  for (long t=0;t<60000000;t++){ s ^= s<<13; s ^= s>>7; s ^= s<<17; double u = (s>>11) * 0x1p-53; double x = -760.0 + u*780.0; if (t%3==0) x = -u*60.0;
    double a=exp(x), m=myexp(x); n++; if (asuint64(a)!=asuint64(m)){ if(bad<5) printf("x=%a exp=%a mine=%a\n",x,a,m); bad++; } }
  printf("tested %ld bad %ld\n",n,bad); return 0; }
