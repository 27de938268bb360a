// Sequence addressing shared by the attention kernels: a "sequence" is the list of token rows one softmax runs
// over -- T frames of one (clip, slot) for temporal attention, S slots of one (clip, frame) for spatial attention.
#pragma once
#include "common.h"

constexpr int ATT_HD = 64;    // head dim (all reference geometries: 768/12, 896/14, 1024/16)

struct SeqDesc {
    int n_outer, n_inner;                      // items = n_outer * n_inner
    long outer_stride, inner_stride, offset;   // base_row = outer*outer_stride + inner*inner_stride + offset
    long pos_stride;                           // rows between consecutive sequence positions
    int L;                                     // sequence length
    int diag;                                  // key allowed iff key_pos <= query_pos + diag (large = no mask)
    int heads, D;
};

__device__ __forceinline__ long seq_base(const SeqDesc& s, int item) {
    const int o = item / s.n_inner, i = item - o * s.n_inner;
    return o * s.outer_stride + i * s.inner_stride + s.offset;
}

static inline int diag_from_causal(int ca) {
    // vit.py:93-99: ca in {1,2}: tril(); ca >= 3: tril(diagonal=ca-2); ca <= 0: no mask.
    if (ca <= 0) return 1 << 28;
    return ca <= 2 ? 0 : ca - 2;
}
static inline SeqDesc temporal_desc(const tcow_attn_shape* s) {
    SeqDesc d;
    d.n_outer = s->B; d.n_inner = s->S - 1; d.outer_stride = (long)s->T * s->S; d.inner_stride = 1; d.offset = 1;
    d.pos_stride = s->S; d.L = s->T; d.diag = diag_from_causal(s->causal); d.heads = s->heads; d.D = s->D;
    return d;
}
static inline SeqDesc spatial_desc(const tcow_attn_shape* s) {
    // cls slot takes part iff causal_attention in {0,1} (vit.py:180-186 vs :202-208)
    const int s0 = (s->causal == 0 || s->causal == 1) ? 0 : 1;
    SeqDesc d;
    d.n_outer = s->B * s->T; d.n_inner = 1; d.outer_stride = s->S; d.inner_stride = 0; d.offset = s0;
    d.pos_stride = 1; d.L = s->S - s0; d.diag = 1 << 28; d.heads = s->heads; d.D = s->D;
    return d;
}
