"""CPU: the C-ABI library builds/loads and exports every symbol include/tcow_hip.h declares (no compute calls)."""
import os
import re

from conftest import ROOT


def _declared():
    txt = open(os.path.join(ROOT, 'include', 'tcow_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(tcow_[a-z0-9_]+)\s*\(', txt)))


def test_header_declares_entry_points():
    names = _declared()
    for must in ['tcow_gemm_nt', 'tcow_gemm_tn', 'tcow_layernorm_fwd', 'tcow_layernorm_bwd', 'tcow_attn_temporal_fwd', 'tcow_attn_spatial_fwd',
                 'tcow_attn_temporal_bwd', 'tcow_attn_spatial_bwd', 'tcow_im2col', 'tcow_embed_fwd', 'tcow_upsample_fwd', 'tcow_last_error']:
        assert must in names


def test_library_exports_every_declared_symbol():
    from tcow_amd import _lib
    for fmt in ('bf16', 'fp16'):          # both builds of the same sources: bfloat16 and binary16 storage of the 16-bit mode
        lib = _lib.lib(fmt)               # raises TcowError if the .so is missing: the product has no fallback
        for name in _declared():
            assert hasattr(lib, name), f'{name} declared in tcow_hip.h but not exported by the {fmt} build'
        assert lib.tcow_version() == _lib.ABI_VERSION
        assert isinstance(lib.tcow_last_error(), bytes)
    assert _lib.lib('bf16') is not _lib.lib('fp16') and _lib.lib('bf16').tcow_version() == _lib.lib('fp16').tcow_version()


def test_python_signatures_cover_the_header():
    from tcow_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()
    hdr = open(os.path.join(ROOT, 'include', 'tcow_hip.h')).read()
    assert int(re.search(r'#define\s+TCOW_ABI_VERSION\s+(\d+)', hdr).group(1)) == _lib.ABI_VERSION      # the version the loader insists on (a stale .so is refused at load)
    assert int(re.search(r'#define\s+TCOW_AMAX_SLOTS\s+(\d+)', hdr).group(1)) == __import__('tcow_amd.ops', fromlist=['x']).AMAX_SLOTS


def test_no_oracle_import_in_product():
    """The product path must never route through the oracle (or any CPU fallback)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'tcow_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f
