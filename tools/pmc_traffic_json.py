"""Turn the two rocprofv3 --pmc passes of `bench.py` (FETCH_SIZE, WRITE_SIZE; one counter group per pass, MI355X_MICROARCH.md) into
profiles/rNN_pmc_gemm_nt.json: HBM bytes per launch of the NT-GEMM family, dispatch-weighted over a training step, with the gfx950 x2
correction on FETCH_SIZE calibrated in the same run on a streaming kernel of known byte count (ln_fwd_kernel: one f32 [M, D] read)."""
import collections, csv, glob, hashlib, json, os, re, sys

def load(path, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(path + '/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get('Counter_Name') != counter:
                continue
            name = re.sub(r'\(anonymous namespace\)::', '', row.get('Kernel_Name', ''))
            a = acc[name]; a[0] += float(row.get('Counter_Value', 0) or 0); a[1] += 1
    return acc

fetch, write = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
out_path = sys.argv[3]
M, D = 27090, 768
def mean(acc, pat):
    tot = n = 0
    for k, (v, c) in acc.items():
        if re.search(pat, k): tot += v; n += c
    return (tot / n if n else None), n
# calibration: known streams (f32 [M, D] read = 83.2 MB; bf16 [M, D] write = 41.6 MB)
sc_f, _ = mean(fetch, r'scale_cast_kernel'); ln_f, _ = mean(fetch, r'ln_fwd_kernel'); sc_w, _ = mean(write, r'scale_cast_kernel')
known_read = M * D * 4 / 1024.0; known_write = M * D * 2 / 1024.0
corr_f = [known_read / x for x in (ln_f,) if x]        # (ln_fwd reads exactly one f32 [M, D] matrix; scale_cast runs on other shapes since round 3)
corr = sum(corr_f) / len(corr_f) if corr_f else 2.0
f_nt, n_f = mean(fetch, r'gemm_nt_bf16'); w_nt, n_w = mean(write, r'gemm_nt_bf16')
per = {}
for k, (v, c) in fetch.items():
    if 'gemm_nt_bf16' in k:
        m = re.search(r'EpiCfg<(-?\d+), (-?\d+)>', k)
        kn = re.match(r'(?:void )?(gemm_nt_bf16\w*)', k)                  # the kernel's own name stays in the key: the 320 and the c2 kernel share epilogue tags
        tag = (kn.group(1) if kn else k[:40]) + (f'<{m.group(1)},{m.group(2)}>' if m else '')
        per[tag] = dict(fetch_kib=v / c, dispatches=c, write_kib=(write[k][0] / write[k][1] if k in write and write[k][1] else None))
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sha = hashlib.sha256(b''.join(open(os.path.join(root, 'tcow_amd', 'csrc', f), 'rb').read() for f in ('gemm_bf16.hip', 'gemm_nt_common.h'))).hexdigest()     # (as bench.py pmc_traffic)
rec = dict(kernel='gemm_nt_bf16_*_kernel (every NT-GEMM launch of the timed steps at M = 27090, dispatch-weighted)',
           command='rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-config-legs (tools/pmc_bench.sh)',
           dispatches_fetch=n_f, dispatches_write=n_w, fetch_size_kib_mean=f_nt, write_size_kib_mean=w_nt,
           fetch_correction=corr, calibration=dict(scale_cast_fetch_kib=sc_f, ln_fwd_fetch_kib=ln_f, known_f32_read_kib=known_read, scale_cast_write_kib=sc_w, known_bf16_write_kib=known_write),
           traffic_bytes_per_launch=(f_nt * corr + w_nt) * 1024.0 if f_nt and w_nt else None, gemm_bf16_sha256=sha, per_kernel=per)
json.dump(rec, open(out_path, 'w'), indent=1)
print(json.dumps({k: rec[k] for k in ('fetch_size_kib_mean', 'write_size_kib_mean', 'fetch_correction', 'traffic_bytes_per_launch')}))
