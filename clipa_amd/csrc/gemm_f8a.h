// Interface between clipa_gemm_nt_f8 (gemm_f8.hip) and the four-wave fp8 kernel (gemm_f8a.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace clipa_gemm {

struct F8AArgs {
  const char* A; const char* B; char* C; char* C2; const float* bias; const char* aux;
  const float* sa; const float* sb;     // de-quantisation scales: per row of A [M], per row of B [N]; NULL = 1
  int M, N, K;                          // K in bytes = fp8 k-values
  long lda, ldb, ldc, ldaux;            // A, B: bytes; C, aux: elements
  float alpha;
  int epi, act, abl, gm;
  const float* so;                      // OUTQ: per-row output scale [M] (the e4m3 output of row m is the bf16 result times so[m]); NULL = 1
  float* cs_part;                       // OUTQ == 2 (DACT): [M / 128][N] partial column sums of the unscaled outputs
  int outq;                             // C holds e4m3 bytes (1 byte per element, row stride ldc BYTES)
  const float* emit_t;                  // CLIPA_EPI_DACT8 with the activation operand of the weight gradient as second output (round 6): C2 = uint8 [M, ldc],
                                        // C2[m, n] = e4m3(act(aux[m, n]) * sa[m] / emit_t[0]) - what clipa_scale_quantize_rows_e4m3 writes; NULL = no such output
  int pre8, aux8;                       // C2 / aux hold e4m3 bytes (1 byte per element, row strides ldc / ldaux in BYTES): CLIPA_EPI_ACT_PRE8 / CLIPA_EPI_DACT8
};

bool f8a_eligible(long M, long N, long K, int fmt_b);
int f8a_launch(const F8AArgs& a, int fmt_a, int dev, int num_cu, hipStream_t st);

}  // namespace clipa_gemm
