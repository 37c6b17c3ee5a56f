// The attention kernels of attention.hip for the wide heads of the reference's largest image towers (open_clip/model_configs:
// ViT-g-14 head dim 88, ViT-bigG-14 104, ViT-e-14 112 - CLIPA-v2's G/14 is the second), non-causal only.  Same source, own
// translation unit (the build compiles its sources in parallel); entry point clipa_attn_wide_launch, called by
// clipa_attention_fwd / clipa_attention_bwd.
#define CLIPA_ATTN_WIDE 1
#include "attention.hip"
