"""CPU: bit-exact token / patch index maps against maps pushed through the reference's own einops patterns (G6)."""
import numpy as np

from conftest import load_golden
from oracle import seeker_oracle as so


def test_token_index_map():
    _, g = load_golden('g6_index_maps')
    for key in [k for k in g if k.startswith('token_src_')]:
        B, T, Hp, Wp = map(int, key.split('_')[2:])
        N = Hp * Wp
        src = g[key]                                   # [b, position-1] -> flat (b*T + t)*N + n of the source token
        for b in range(B):
            for t in range(T):
                for n in (0, 1, N // 2, N - 1):
                    assert src[b, so.ref_token_index(t, n, T) - 1] == (b * T + t) * N + n
        pos = g[key.replace('token_src', 'token_pos')]  # [b, t, h, w] -> position-1 in the token list
        for b in range(B):
            for t in (0, T - 1):
                for h in (0, Hp - 1):
                    for w in (0, Wp // 2, Wp - 1):
                        assert pos[b, t, h, w] == b * N * T + so.ref_token_index(t, h * Wp + w, T) - 1


def test_unpatchify_map():
    _, g = load_golden('g6_index_maps')
    for key in [k for k in g if k.startswith('unpatchify_')]:
        C, P, Hp, Wp = map(int, key.split('_')[1:])
        m = g[key]                                     # [c, y, x] -> flat index into (H', W', C*P*P)
        for c in range(C):
            for y in range(Hp * P):
                for x in range(Wp * P):
                    n = (y // P) * Wp + (x // P)
                    assert m[c, y, x] == n * C * P * P + so.patch_pixel_index(c, y % P, x % P, P)


def test_patchify_map():
    _, g = load_golden('g6_index_maps')
    pat = g['patchify_4_4_2_3']                        # [n, k] -> flat pixel index in a (C=4, 8, 12) image
    P, C, Hp, Wp, H, W = 4, 4, 2, 3, 8, 12
    for n in range(Hp * Wp):
        for c in range(C):
            for py in range(P):
                for px in range(P):
                    y, x = (n // Wp) * P + py, (n % Wp) * P + px
                    assert pat[n, so.patch_pixel_index(c, py, px, P)] == (c * H + y) * W + x
