/* Exhaustive CPU validation of the two identities the device Halton code relies on (trg_device.h halton_c):
 *  (1) for every prime base b of the table and every integer n < 2^22:  floor((n + 0.5) * fl(1/b)) == n div b
 *      and fma(-q, b, n) == n mod b;
 *  (3) the same quotient identity for every power b^k <= 2^22 (the digits are then independent of one another);
 *  (2) the closed form of the base-2 radical inverse equals the reference's fp32 loop (strided sample here;
 *      the full 2^32 sweep was run once when the code was written: 0 mismatches).
 * Build: gcc -O2 -fopenmp -mfma -ffp-contract=off halton_identities.c -lm */
#include <stdio.h>
#include <stdint.h>
#include <math.h>
static const uint32_t P[64] = {2,3,5,7,11,13,17,19,23,29,31,37,41,43,47,53,59,61,67,71,73,79,83,89,97,101,103,107,109,113,127,131,137,139,149,151,157,163,167,173,179,181,191,193,197,199,211,223,227,229,233,239,241,251,257,263,269,271,277,281,283,293,307,311};
int main(){
  long bad=0;
  for(int d=1; d<64; ++d){
    uint32_t b=P[d]; float bf=(float)b, rcp=1.0f/bf, h=0.5f/bf;
    uint32_t lim = 1u<<22;
    #pragma omp parallel for reduction(+:bad)
    for(uint32_t n=0;n<lim;++n){
      float nf=(float)n;
      float q=floorf((nf+0.5f)*rcp);
      float dg=fmaf(-q,bf,nf);
      if((uint32_t)q!=n/b || (uint32_t)dg!=n%b || dg<0) bad++;
    }
  }
  printf("bad=%ld\n",bad);
  // (3) every quotient straight from n: floor((n + 0.5) * fl(1 / b^k)) == n div b^k for all n < 2^22 and all b^k <= 2^22
  long bad3=0; long powers=0;
  for(int d=1; d<64; ++d){
    uint32_t b=P[d];
    for(uint64_t m=(uint64_t)b*b; m<=(1ull<<22); m*=b){
      float rcp=1.0f/(float)m; uint32_t mm=(uint32_t)m; powers++;
      #pragma omp parallel for reduction(+:bad3)
      for(uint32_t n=0;n<(1u<<22);++n){
        float q=floorf(((float)n+0.5f)*rcp);
        if((uint32_t)q!=n/mm) bad3++;
      }
    }
  }
  printf("bad3=%ld powers=%ld\n",bad3,powers);
  // base-2 closed form vs loop, sampled + structured
  long bad2=0;
  #pragma omp parallel for reduction(+:bad2)
  for(uint64_t t=0;t<(1ull<<32);t+=4099){
    uint32_t i=(uint32_t)t;
    // reference loop
    float f=1.0f, r=0; uint32_t x=i; while(x){ f=f*0.5f; r=r+f*(float)(x&1); x>>=1; }
    uint32_t rev=0; for(int k=0;k<32;++k) if(i>>k&1) rev|=1u<<(31-k);
    uint32_t lz = rev? __builtin_clz(rev):32;
    uint32_t drop = lz<8? 8-lz:0;
    uint32_t kept=(rev>>drop)<<drop;
    float c=(float)kept*0x1p-32f;
    if(drop){ uint32_t tie=(rev>>(drop-1))&1, lsb=(rev>>drop)&1; if(tie&lsb){ union{uint32_t u; float f;} u; u.u=(127-lz-24)<<23; c+=u.f; } }
    if(c!=r) bad2++;
  }
  printf("bad2=%ld\n",bad2);
}
