"""Per-kernel-family roofline table of the bf16 training step from a rocprofv3 kernel-stats summary (tools/prof_summary.py output):
algorithmic work per step (formulas of DESIGN.md section 3, BASELINE configs[1]: M = 27 090 token rows, D = 768, 12 blocks) over the measured
time per step, against the MI355X peaks (2.5 PFLOP/s dense bf16, 8 TB/s HBM).   usage: python tools/roofline_table.py profiles/r02_step_kernel_stats.txt [steps]"""
import re, sys
path = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rows = {}
for l in open(path):
    m = re.match(r'(.+?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$', l)
    if m:
        rows[m.group(1).strip()] = (int(m.group(2)) / steps, float(m.group(3)) / steps)
def fam(*pats):
    c = t = 0.0
    for k, (cc, tt) in rows.items():
        if any(p in k for p in pats):
            c += cc; t += tt
    return c, t
M, D, L, heads, T, S, BQ = 27090, 768, 12, 12, 30, 301, 3
gemm_fwd = 2.0 * M * (16 * D * D) * L                       # qkv x2, folded temporal proj . temporal_fc, spatial proj, fc1, fc2 per block
head = 2.0 * M * D * D
nt_flops = 2 * gemm_fwd + 2 * head                           # forward + input gradients (+ mask head both ways)
tn_flops = gemm_fwd + head
sp_f = 4.0 * S * S * 64 * (BQ * T) * heads * L; tp_f = 4.0 * 32 * 32 * 64 * (BQ * (S - 1)) * heads * L      # attention score + value products (padded temporal tile)
qkv_b = M * 3 * D * 2.0; o_b = M * D * 2.0
PF, TB = 2500.0, 8.0
out = []
def row(name, launches, us, flops=None, bytes_=None, note=''):
    if us <= 0:
        return
    tf = flops / us / 1e6 if flops else None; tb = bytes_ / us / 1e6 if bytes_ else None
    out.append(f'| {name} | {launches:.0f} | {us / 1e3:.2f} | ' + (f'{flops / 1e12:.2f} TF' if flops else '-') + ' | ' + (f'{bytes_ / 1e9:.2f} GB' if bytes_ else '-') + ' | ' +
               (f'{tf:.0f} TFLOP/s = {tf / PF:.3f}' if tf else '-') + ' | ' + (f'{tb:.2f} TB/s = {tb / TB:.2f}' if tb else '-') + f' | {note} |')
c, t = fam('gemm_nt_bf16'); row('NT GEMM (forward + input gradients, fused epilogues)', c, t, nt_flops, None, 'T = main loop (1.3-1.4 PFLOP/s inside it) + epilogue bytes at 5-6 TB/s, the two do not overlap on a CU (persistent loop, two tiles per CU, packed-f32 arithmetic all measured: DESIGN.md 3); at or above hipBLASLt on every shape (profiles/r04_vendor_gemm.txt)')
c, t = fam('gemm_tn_bf16'); c2, t2 = fam('slab_reduce4'); row('weight-gradient GEMM (grouped) + folds', c + c2, t + t2, tn_flops, None, 'a real-data MFMA stream tops out at 1.8-1.9 PFLOP/s (power-managed clock, profiles/r04_ubench_mfma.txt); loads + barriers 14 %, slab store + column sums 16 %')
c, t = fam('attn_fwd_stream'); row('spatial attention forward', c, t, sp_f, L * (qkv_b + o_b), 'VALU-issue / latency-bound at d = 64 (62 VALU + 16 v_exp_f32 per 8 MFMAs, 4 waves per SIMD; profiles/r05_attn_fwd_p4.txt, r06_attn_pack_half.txt), traffic at the algorithmic minimum')
c, t = fam('attn_bwd_one_kernel', 'attn_bwd_dq_stream', 'attn_bwd_dkv_stream')
row('spatial attention backward (one kernel: dK/dV tiles + dQ^T from the dS strip in LDS)', c, t, 2.5 * sp_f, L * (2 * qkv_b + 2 * o_b), 'VALU / LDS-traffic bound tile steps (10 tile waves + 2 dQ chain waves per CU); prologue at the one-CU miss rate')
c, t = fam('attn_fwd_mfma<false>'); row('temporal attention forward', c, t, tp_f, L * (qkv_b + o_b), 'HBM-bound (81 % of a float4 copy)')
c, t = fam('attn_bwd_one_tile'); row('temporal attention backward', c, t, 2.5 * tp_f, L * (2 * qkv_b + o_b), 'HBM-bound (84 % of a float4 copy)')
c, t = fam('ln_fwd_kernel'); row('LayerNorm forward', c, t, None, c * M * D * 6.0, 'HBM-bound')
c, t = fam('ln_bwd_kernel'); row('LayerNorm backward (+ residual add, + 16-bit copy of dx for the next GEMM)', c, t, None, c * M * D * (2 + 4 + 4 + 4 + 2.0), 'HBM-bound')
c, t = fam('adamw_kernel', 'adamw_cast_kernel', 'sumsq_kernel'); row('clip + AdamW (the tile kernel also writes the 16-bit W / W^T copies of the GEMM weights)', c, t, None, 122.1e6 * 32.0 + 107.0e6 * 4.0, 'HBM-bound')
c, t = fam('cast_transpose_batched'); row('re-cast of the folded products (bf16 W\' and W\'^T)', c, t, None, 12 * 768 * 768 * 8.0, 'latency (7 MB)')
total = sum(tt for _, tt in rows.values())
print(f'Roofline table of the bf16 training step (BASELINE configs[1], {steps} profiled steps, {total / 1e3:.2f} ms of kernels per step; source: {path}).')
print('Peaks: 2.5 PFLOP/s dense bf16 MFMA, 8 TB/s HBM3E (about 6.3 TB/s is achievable by a streaming kernel).\n')
print('| kernel family | launches / step | ms / step | algorithmic FLOPs / step | algorithmic bytes / step | achieved vs MFMA peak | achieved vs HBM peak | bound |')
print('|---|---|---|---|---|---|---|---|')
print('\n'.join(out))
listed = sum(float(o.split('|')[3]) for o in out)
print(f'\nListed families: {listed:.2f} ms of {total / 1e3:.2f} ms; the rest is the mask objective, mask builder, elementwise glue and tensor-op launches.')
