#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
/* sensitivity of the feedback modulator: two copies, exact arithmetic, one perturbed by eps in phase at n = 0 */
int main(int argc,char**argv){
  double beta_d=argc>1?atof(argv[1]):0.4; int B=argc>2?atoi(argv[2]):1; double val=argc>3?atof(argv[3]):0.0; float beta=(float)beta_d; const double sr=48000.0; double eps=1e-12;
  double p=0.1,q=0.1+eps; float ring_p[2048]={0}, ring_q[2048]={0};
  for(long n=0;n<60L*48000;n++){
    float fp=ring_p[n%B]*beta, fq=ring_q[n%B]*beta;
    double dp=440.0*pow(2.0,(double)fp+val)/sr, dq=440.0*pow(2.0,(double)fq+val)/sr;
    /* the f32 cast of the sine hides tiny perturbations (quantisation); keep f64 here to see the linear dynamics */
    double sp=sin(p*M_PI*2.0), sq=sin(q*M_PI*2.0); ring_p[n%B]=(float)sp; ring_q[n%B]=(float)sq;
    p=fmod(p+dp,1.0); q=fmod(q+dq,1.0);
    if(n%(5*48000)==0||n==60L*48000-1){ double d=q-p; if(d>0.5)d-=1; if(d<-0.5)d+=1; printf("t %5.1f s  phase difference %.3e (x %.2f of eps)\n",n/48000.0,d,d/eps); }
  }
}
