// Toolchain / ISA-semantics probe for gfx950 (not product code).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short s4;
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(16))) float f16v;

__device__ inline short f2bf(float f){ unsigned u=__float_as_uint(f); u=(u+0x7FFF+((u>>16)&1))>>16; return (short)u; }

// C[16x16] = A[16x32] * B[32x16]; A row-major [16][32], B row-major [32][16] (bf16 bits)
extern "C" __global__ void k_mfma16(const short* A, const short* B, float* C){
  int l=threadIdx.x; int r=l&15, g=l>>4;
  bf16x8 a,b;
  for(int j=0;j<8;j++){ a[j]=A[r*32+8*g+j]; b[j]=B[(8*g+j)*16+r]; }
  f4 acc={0,0,0,0};
  acc=__builtin_amdgcn_mfma_f32_16x16x32_bf16(a,b,acc,0,0,0);
  for(int i=0;i<4;i++) C[(4*g+i)*16+r]=acc[i];   // row=4*(lane>>4)+reg, col=lane&15
}
// C[32x32] = A[32x16]*B[16x32]
extern "C" __global__ void k_mfma32(const short* A, const short* B, float* C){
  int l=threadIdx.x; int r=l&31, h=l>>5;
  bf16x8 a,b;
  for(int j=0;j<8;j++){ a[j]=A[r*16+8*h+j]; b[j]=B[(8*h+j)*32+r]; }
  f16v acc; for(int i=0;i<16;i++) acc[i]=0;
  acc=__builtin_amdgcn_mfma_f32_32x32x16_bf16(a,b,acc,0,0,0);
  for(int i=0;i<16;i++) C[((i&3)+8*(i>>2)+4*h)*32+r]=acc[i];
}
// tr read: LDS image [rows=16][cols=64] shorts, value = row*64+col. each 16-lane group g reads block
// rows 4g'..: lane 4q+p supplies address of row (r0+q), cols 4p..4p+3
extern "C" __global__ void k_tr(short* out){
  __shared__ __attribute__((aligned(16))) short lds[16*64];
  int l=threadIdx.x;
  for(int i=l;i<16*64;i+=64) lds[i]=(short)i;
  __syncthreads();
  int g=l>>4, i16=l&15, q=i16>>2, p=i16&3;
  int row=4*g+q, col=4*p;   // group g reads rows 4g..4g+3, cols 0..15
  s4 v=__builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds+row*64+col));
  for(int j=0;j<4;j++) out[l*4+j]=v[j];
}
// simple axpy to test stream interop
extern "C" __global__ void k_fill(float* p, int n, float v){ int i=blockIdx.x*blockDim.x+threadIdx.x; if(i<n) p[i]=v+i; }

extern "C" int probe_fill(float* p, int n, float v, void* stream){
  hipLaunchKernelGGL(k_fill, dim3((n+255)/256), dim3(256), 0, (hipStream_t)stream, p, n, v);
  return (int)hipGetLastError();
}
extern "C" int probe_mfma16(const short* A,const short* B,float* C,void* s){ hipLaunchKernelGGL(k_mfma16,dim3(1),dim3(64),0,(hipStream_t)s,A,B,C); return (int)hipGetLastError(); }
extern "C" int probe_mfma32(const short* A,const short* B,float* C,void* s){ hipLaunchKernelGGL(k_mfma32,dim3(1),dim3(64),0,(hipStream_t)s,A,B,C); return (int)hipGetLastError(); }
extern "C" int probe_tr(short* o,void* s){ hipLaunchKernelGGL(k_tr,dim3(1),dim3(64),0,(hipStream_t)s,o); return (int)hipGetLastError(); }
