"""Algorithmic FLOP / byte counts of the Seeker hot path, used by bench.py's roofline block."""


def gemm_flops(M, N, K):
    return 2.0 * M * N * K


def seeker_forward_flops(B, T, Hp, Wp, D, heads, depth, P=16, Ci=4, Co=3, F=3, mlp_ratio=4):
    """FLOPs of one query forward with the reference's token counts (one cls per clip, L = N*T patch tokens)."""
    N = Hp * Wp; L = N * T; S = N + 1; d = D // heads
    per_block = (2.0 * B * L * D * 3 * D            # temporal qkv
                 + B * N * heads * 4.0 * T * T * d  # temporal QK^T + AV
                 + 2 * 2.0 * B * L * D * D          # temporal proj + temporal_fc
                 + 2.0 * B * T * S * D * 3 * D      # spatial qkv
                 + B * T * heads * 4.0 * S * S * d  # spatial QK^T + AV
                 + 2.0 * B * T * S * D * D          # spatial proj
                 + 2 * 2.0 * B * (L + 1) * D * mlp_ratio * D)   # mlp
    attn = depth * (B * N * heads * 4.0 * T * T * d + B * T * heads * 4.0 * S * S * d)
    total = depth * per_block + 2.0 * B * L * (Ci * P * P) * D + 2.0 * B * L * D * (Co * P * P + F)
    return dict(total=total, attention=attn, per_block=per_block)


def attention_bytes(B, T, S, D, elem=2):
    """Algorithmic HBM bytes of one attention launch: read Q, K, V and write O once."""
    return 4.0 * B * T * S * D * elem
