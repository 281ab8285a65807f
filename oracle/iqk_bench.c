/* iqk_bench.c — test infrastructure (bench.py's CPU leg of the q4_k_m workload only): drives the reference's OWN llamafile
 * GEMM kernels (third_party/llamafile/iqk_mul_mat.inc, compiled unmodified into oracle/_ref/libiqk_ref_*.so) through the
 * bs=1 control flow of LLAMA_MOE_TP::forward_one (kt-kernel/operators/llamafile/moe.hpp:271-460): x -> Q8_K; per routed
 * expert gate / up GEMVs (m_block rows x 1 column each there, iqk's ith / nth row partition here), fp32 silu(gate) * up,
 * -> Q8_K, down GEMV.  OpenMP threads, barriers between the stages.  The function pointer is resolved by the Python side
 * (dlsym), so this file needs no reference header.  Built into libktx_oracle.so. */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>

typedef _Bool (*iqk_fn)(long, long, long, int, const void*, long, int, const void*, long, float*, long, int, int);
void ktxo_quantize_row_q8_K(const float* x, int K, int8_t* q, float* d, int16_t* bsums);   /* ktx_oracle_gguf.c */

typedef struct { float d; int8_t qs[256]; int16_t bsums[16]; } blk_q8k;

static void to_q8k(const float* x, int K, blk_q8k* out, int8_t* q, float* d, int16_t* bs) {
  ktxo_quantize_row_q8_K(x, K, q, d, bs);
  for (int b = 0; b < K / 256; b++) {
    out[b].d = d[b];
    for (int i = 0; i < 256; i++) out[b].qs[i] = q[b * 256 + i];
    for (int i = 0; i < 16; i++) out[b].bsums[i] = bs[b * 16 + i];
  }
}

/* `iters` layer forwards; weights[(set * k + e) * 3 + {0,1,2}] = gate / up / down blocks of expert e of layer set `set`.
 * Returns seconds. */
double ktxo_iqk_bench(void* fnp, int H, int I, int k, const int* types, const void** weights, int nsets, const float* x,
                      int nthreads, int iters) {
  iqk_fn fn = (iqk_fn)fnp;
  const int KM = H > I ? H : I;
  blk_q8k* xq = (blk_q8k*)malloc(sizeof(blk_q8k) * (H / 256));
  blk_q8k* aq = (blk_q8k*)malloc(sizeof(blk_q8k) * (I / 256));
  int8_t* q = (int8_t*)malloc(KM);
  float* d = (float*)malloc(sizeof(float) * (KM / 256));
  int16_t* bs = (int16_t*)malloc(sizeof(int16_t) * (KM / 16));
  float* g = (float*)malloc(sizeof(float) * I);
  float* u = (float*)malloc(sizeof(float) * I);
  float* a = (float*)malloc(sizeof(float) * I);
  float* y = (float*)malloc(sizeof(float) * H);
  float* acc = (float*)calloc(H, sizeof(float));
  const double t0 = omp_get_wtime();
#pragma omp parallel num_threads(nthreads)
  {
    const int ith = omp_get_thread_num(), nth = omp_get_num_threads();
    for (int it = 0; it < iters; it++) {
      const void** w = weights + (size_t)(it % nsets) * k * 3;
#pragma omp single
      to_q8k(x, H, xq, q, d, bs);
      for (int e = 0; e < k; e++) {
        fn(I, 1, H, types[0], w[e * 3 + 0], H / 256, 15, xq, H / 256, g, I, ith, nth);
        fn(I, 1, H, types[1], w[e * 3 + 1], H / 256, 15, xq, H / 256, u, I, ith, nth);
#pragma omp barrier
#pragma omp for
        for (int i = 0; i < I; i++) a[i] = g[i] / (1.0f + expf(-g[i])) * u[i];
#pragma omp single
        to_q8k(a, I, aq, q, d, bs);
        fn(H, 1, I, types[2], w[e * 3 + 2], I / 256, 15, aq, I / 256, y, H, ith, nth);
#pragma omp barrier
#pragma omp for
        for (int i = 0; i < H; i++) acc[i] += 0.5f * y[i];
      }
    }
  }
  const double dt = omp_get_wtime() - t0;
  free(xq); free(aq); free(q); free(d); free(bs); free(g); free(u); free(a); free(y); free(acc);
  return dt;
}
