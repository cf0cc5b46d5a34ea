/* Proof by exhaustion behind the fp16 fast quotient of sweeps 2 / 3 (vidcom2_amd/csrc/vc2_kernels.hip, k_norm_colsum2):
 *     r = approx(1 / dn);  q0 = x * r;  e = fma(-dn, q0, x);  q = fma(e, r, q0)
 * equals the IEEE fp32 quotient x / dn -- what torch's CPU fp16 division computes before it rounds to fp16
 * (reference token_compressor/vidcom2/vidcom2.py:48, F.normalize) -- for EVERY nonzero finite fp16 x and dn (subnormals
 * included), with r anywhere within +-4 fp32 ulps of the correctly rounded reciprocal (v_rcp_f32 is accurate to 1 ulp).
 * 9 x 1.0e9 cases, ~20 s on 8 threads:   gcc -O2 -fopenmp -ffp-contract=off check_f16_quotient.c -o chk -lm && ./chk
 * (tests/test_abi_and_host.py runs a strided subset.)  Prints one line per reciprocal offset; exit code 1 on any mismatch.
 * Usage: chk [stride]   -- stride > 1 visits every stride-th denominator. */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <math.h>
#include <string.h>
static float h2f(uint16_t h){ uint32_t s=(h>>15)&1,e=(h>>10)&31,m=h&1023; float v; if(e==0) v=ldexpf((float)m,-24); else if(e==31) v=m?NAN:INFINITY; else v=ldexpf((float)(m+1024),e-25); return s?-v:v; }
int main(int argc, char** argv){ int stride = argc > 1 ? atoi(argv[1]) : 1; if (stride < 1) stride = 1;
  long bad[9]={0}; 
  #pragma omp parallel for schedule(dynamic,64)
  for(int d=0x0001; d<0x7C00; d += stride){ float dn=h2f(d); float rc=1.0f/dn; long lb[9]={0};
    for(int v=0; v<9; ++v){ uint32_t rb; memcpy(&rb,&rc,4); rb += (v-4); float r; memcpy(&r,&rb,4);
    for(int xh=0x0001; xh<0x7C00; ++xh){ float x=h2f(xh); float qr=x/dn; 
      float q0=x*r; float e=fmaf(-dn,q0,x); float q=fmaf(e,r,q0);
      uint32_t a,b; memcpy(&a,&q,4); memcpy(&b,&qr,4);
      if(a!=b) lb[v]++;
    } }
    #pragma omp critical
    { for(int v=0;v<9;++v){bad[v]+=lb[v];} }
  }
  long all=0; for(int v=0;v<9;++v){ printf("r off by %+d ulp: f32 mismatches %ld\n",v-4,bad[v]); all+=bad[v]; } return all ? 1 : 0; }
