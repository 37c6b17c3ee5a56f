// Streaming accesses for the HBM-bound kernels of clipa_amd (gfx950).
// Non-temporal 8 / 16-byte loads and stores for operands a kernel touches exactly once (row streams of the LayerNorm /
// quantise / cast / re-materialise kernels): +4 % on a one-shot grid (tools/probes/stream_ab.hip).
#pragma once
#include "common.h"

template <class V> __device__ __forceinline__ V ld_stream(const void* p) { return __builtin_nontemporal_load((const V*)p); }
template <class V> __device__ __forceinline__ void st_stream(void* p, V v) { __builtin_nontemporal_store(v, (V*)p); }
