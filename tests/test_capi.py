"""CPU: the C-ABI shared library builds for gfx950, loads, and exports every symbol include/sjd_hip.h declares."""
import ctypes
import os
import re

from tests.conftest import ROOT


def _ensure_built():
    import sjd_amd._lib as L
    if not os.path.exists(L.SO_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return L


_DECL = re.compile(r"^(?:int|int64_t|uint64_t|void \*|void|float|const char \*)\s*(sjd_[a-z0-9_]+)\s*\(", re.M)      # a declaration, not a mention in a comment


def _nm_exports(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return {ln.split()[-1] for ln in out.splitlines() if " T sjd_" in ln and not ln.split()[-1].startswith("sjd_debug_trace")}


def test_exports_match_header():
    """the PRODUCT library: include/sjd_hip.h == _lib.EXPORTS == what libsjd_hip.so exports (nm), and nothing experimental among it"""
    L = _ensure_built()
    lib = L.load()
    hdr = open(os.path.join(ROOT, "include", "sjd_hip.h")).read()
    declared = set(_DECL.findall(hdr))
    assert declared == set(L.EXPORTS), declared ^ set(L.EXPORTS)
    assert _nm_exports(L.SO_PATH) == declared, _nm_exports(L.SO_PATH) ^ declared
    assert len(declared) <= 45 and not (declared & set(L.EXP_EXPORTS))                            # the experiments live elsewhere
    for name in declared:
        assert getattr(lib, name) is not None
    for name in L.EXP_EXPORTS:
        assert not hasattr(lib, name), name
    assert lib.sjd_version() == int(re.search(r"#define SJD_VERSION (\d+)", hdr).group(1))
    assert lib.sjd_error_string(-2) == b"unsupported configuration"


def test_experimental_library_is_a_superset_with_its_own_header():
    """libsjd_hip_exp.so (same sources, -DSJD_EXPERIMENTAL): everything the product exports plus exactly what include/sjd_hip_experimental.h declares"""
    L = _ensure_built()
    exp = L.load_exp()
    hdr = open(os.path.join(ROOT, "include", "sjd_hip_experimental.h")).read()
    declared = set(_DECL.findall(hdr))
    assert declared == set(L.EXP_EXPORTS), declared ^ set(L.EXP_EXPORTS)
    assert _nm_exports(L.EXP_SO_PATH) == declared | set(L.EXPORTS)
    for name in declared | set(L.EXPORTS):
        assert getattr(exp, name) is not None
    assert exp.sjd_mlp_pair_z(None, None, None, 32, 1, None, None, None, 32, 0, None, 32, 11008, 4096, 768, None, None, 256, None) == -1


def test_product_package_does_not_load_the_experimental_library():
    """importing the product modules and building an engine's host side must not pull libsjd_hip_exp.so in (a fresh interpreter: this process
    has loaded it in the test above)"""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import sjd_amd, sjd_amd._lib as L, sjd_amd.ops, sjd_amd.engine as E, sjd_amd.engine_batch, sjd_amd.backbones; "
            "L.load(); E.check_reduce_timeouts(); assert L._exp is None; "
            "assert 'libsjd_hip_exp' not in open('/proc/self/maps').read(); print('ok')" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


def test_struct_layouts_match_header():
    L = _ensure_built()
    assert ctypes.sizeof(L.RowRule) == 52 and L.RowRule.temperature.offset == 48
    W = 64                                                                                        # SJD_MAX_WINDOW (include/sjd_hip.h)
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "sjd_hip.h")).read()
    assert f"#define SJD_MAX_WINDOW {W} " in src and L.MAX_WINDOW == W
    assert ctypes.sizeof(L.IterParams) == 64 + 8 * W + 2 * 52 * W                                 # static_assert'ed in sjd_sampling.hip
    assert L.IterParams.kv_len.offset == 4 and L.IterParams.batch_rows.offset == 0x14            # the inline s_load offsets of sjdi_kv_rows
    assert L.IterParams.philox_blocks.offset == 28 and L.IterParams.philox_seed.offset == 32 and L.IterParams.philox_offset.offset == 40
    assert L.IterParams.fresh_tok.offset == 64 and L.IterParams.rules.offset == 64 + 8 * W
    assert ctypes.sizeof(L.State) == 16 + 8 * W * 2 + 4 * W + 8 * W
    assert L.State.tokens.offset == 16 and L.State.win_tok.offset == 16 + 8 * W and L.State.q_src.offset == 16 + 16 * W
    assert L.State.amax.offset == 16 + 16 * W + 4 * W
    assert ctypes.sizeof(L.HeadPartials) == 96 and L.HeadPartials.row_sumsq.offset == 48          # static_assert'ed in sjd_sampling.hip
    assert L.HeadPartials.zero_state.offset == 88                                                 # (round 4: the rows' zero state, last member)
    assert ctypes.sizeof(L.RowNorm) == 24


def test_bad_arguments_are_rejected_without_a_gpu():
    L = _ensure_built()
    lib = L.load()
    assert lib.sjd_reguess(None, None, None, 1, 16, None) == -1
    assert lib.sjd_logits_to_probs_sample(None, None, 0, 1.0, 16, 100, None, None, None, None, None) == -1
    assert lib.sjd_verify_accept(None, None, None, None, None, None, None, 16, 100, None) == -1
    assert lib.sjd_attention_workspace_bytes(2, 32, 16, 128, 8) == 2 * 32 * 1 * 8 * 16 * 130 * 4
    # round 4 entry points: argument checks run before any launch
    assert lib.sjd_draft_window_attention_colsplit(None, None, None, None, 2, 16, 32, 32, 128, 1024, 0, None, None, 0, None) == -1
    assert lib.sjd_draft_window_attention_fp8_colsplit(None, None, None, None, 2, 16, 32, 32, 128, 1024, 0, 1.0, 1.0, None, None, 0, None) == -1


def test_host_wait_on_the_mirror_sequence_word():
    """sjd_host_wait_u64 is plain host code (the spin on the word K4 publishes behind the mirrored state): it returns at once when the word
    already matches, when another thread stores the value, and reports a timeout otherwise; bad arguments are rejected."""
    import threading
    import time
    L = _ensure_built()
    lib = L.load()
    word = ctypes.c_uint64(41)
    addr = ctypes.addressof(word)
    assert lib.sjd_host_wait_u64(addr, 41, 1000) == 0
    t0 = time.perf_counter()
    assert lib.sjd_host_wait_u64(addr, 42, 20000) == -3          # SJD_ERR_LAUNCH after ~20 ms
    assert 0.015 < time.perf_counter() - t0 < 2.0
    threading.Timer(0.05, lambda: setattr(word, "value", 42)).start()
    assert lib.sjd_host_wait_u64(addr, 42, 5_000_000) == 0       # ctypes released the GIL: the timer thread could run
    assert lib.sjd_host_wait_u64(None, 1, 10) == -1
    assert lib.sjd_gateup_silu(None, None, None, 32, 11008, 4096, 0, 0, None, None) == -1
    assert lib.sjd_upload_async(None, None, 16, None) == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    import sjd_amd._lib as L
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "SO_PATH", str(tmp_path / "nope.so"))
    try:
        L.load()
        raise AssertionError("expected SjdLibraryError")
    except L.SjdLibraryError as e:
        assert "no CPU/torch fallback" in str(e)
