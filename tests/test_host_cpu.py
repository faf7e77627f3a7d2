"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol
the header declares, the VAE surface matches the reference's (keys, shapes, init,
pickle path), and the product path never routes through the oracle."""
import io
import json
import os
import re

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from conftest import GOLDEN, REPO  # noqa: E402


def _header_symbols(name="rawvae_hip.h"):
    with open(os.path.join(REPO, "include", name)) as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rv_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import ctypes
    from rawaudiovae_kelsey_amd import _lib
    lib = _lib.lib()
    names = _header_symbols()
    diag = _header_symbols("rawvae_hip_diag.h")      # the test hook: exported, not part of the product ABI
    assert 30 <= len(names) <= 70 and diag == ["rv_gemm_force_tile", "rv_plan_diag_skip"]
    raw = ctypes.CDLL(_lib.LIB_PATH)
    # ... and nothing else: launchers only the step plan calls (csrc/internal.h) are hidden, not a second ABI
    import subprocess
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True)
    if nm.returncode == 0:
        dyn = sorted(ln.split()[-1] for ln in nm.stdout.splitlines() if " T rv_" in ln)
        assert dyn == sorted(names + diag), set(dyn) ^ set(names + diag)
    for n in names + diag:
        assert hasattr(raw, n), "header declares %s but the library does not export it" % n
        assert n in _lib.EXPORTED, "%s has no ctypes signature in _lib.py" % n
    assert set(_lib.EXPORTED) <= set(names) | set(diag), set(_lib.EXPORTED) - set(names) - set(diag)
    assert lib.rv_version() >= 100


def test_pad_dims_and_errors():
    from rawaudiovae_kelsey_amd import _lib
    assert _lib.pad_dims(4096, 1024, 2048, 64) == (4096, 1024, 2048, 64)
    assert _lib.pad_dims(32, 512, 2048, 8) == (128, 512, 2048, 64)
    assert _lib.pad_dims(2137, 1000, 96, 100) == (2176, 1024, 128, 128)
    assert _lib.pad_dims(1, 1, 1, 256)[3] == 256
    with pytest.raises(_lib.RvError):
        _lib.pad_dims(0, 1, 1, 1)
    with pytest.raises(_lib.RvError):
        _lib.pad_dims(1, 1, 1, 257)


def test_vae_surface_matches_reference():
    from rawvae.model import VAE, loss_function  # the reference's import path (train.py:11)
    m = VAE(1024, 2048, 256)
    assert (m.segment_length, m.n_units, m.latent_dim) == (1024, 2048, 256)
    sd = m.state_dict()
    assert list(sd) == ["fc1.weight", "fc1.bias", "fc21.weight", "fc21.bias", "fc22.weight", "fc22.bias",
                        "fc3.weight", "fc3.bias", "fc4.weight", "fc4.bias"]
    assert sd["fc1.weight"].shape == (2048, 1024) and sd["fc21.weight"].shape == (256, 2048)
    assert sd["fc3.weight"].shape == (2048, 256) and sd["fc4.weight"].shape == (1024, 2048)
    assert sum(p.numel() for p in m.parameters()) == 5772800  # default.ini model size
    assert all(v.dtype == torch.float32 for v in sd.values())
    for name in ("encode", "reparameterize", "decode", "forward"):
        assert callable(getattr(m, name))
    assert callable(loss_function)


def test_default_init_is_bitwise_the_references():
    """Same construction order + nn.Linear default init => same tensors under the same seed
    (fixture: statistics of the reference's VAE(64,96,8) under torch.manual_seed(0))."""
    from rawvae.model import VAE
    with open(os.path.join(GOLDEN, "summary.json")) as f:
        st = json.load(f)["init_seed0_64_96_8"]
    torch.manual_seed(0)
    m = VAE(64, 96, 8)
    for k, v in m.state_dict().items():
        a = v.numpy().astype(np.float64)
        assert a.min() == st[k]["min"] and a.max() == st[k]["max"], k
        assert abs(a.sum() - st[k]["sum"]) < 1e-9 and list(a.reshape(-1)[:4]) == st[k]["first"], k
        assert np.abs(a).max() <= st[k]["bound"]


def test_pickle_path_and_state_dict_roundtrip():
    from rawvae.model import VAE
    m = VAE(64, 96, 8)
    buf = io.BytesIO()
    torch.save(m, buf)  # train.py:244 saves the whole module
    buf.seek(0)
    m2 = torch.load(buf, weights_only=False)
    assert type(m2).__module__ == "rawvae.model" and type(m2).__name__ == "VAE"
    state = {"epoch": 3, "state_dict": m.state_dict(),
             "optimizer": torch.optim.Adam(m.parameters(), lr=1e-4).state_dict()}  # train.py:208-212
    buf = io.BytesIO()
    torch.save(state, buf)
    buf.seek(0)
    st = torch.load(buf, weights_only=False)
    m3 = VAE(64, 96, 8)
    m3.load_state_dict(st["state_dict"])
    for a, b in zip(m.parameters(), m3.parameters()):
        assert torch.equal(a, b)


def test_cpu_tensors_fail_loudly():
    from rawaudiovae_kelsey_amd._lib import RvError
    from rawvae.model import VAE, loss_function
    m = VAE(64, 96, 8)
    with pytest.raises(RvError):
        m(torch.zeros(2, 64))
    with pytest.raises(RvError):
        loss_function(torch.zeros(2, 64), torch.zeros(2, 64), torch.zeros(2, 8), torch.zeros(2, 8), 1e-4, 64)
    from rawaudiovae_kelsey_amd.engine import TrainEngine
    with pytest.raises(RvError):
        TrainEngine(64, 96, 8, 16, device="cpu")


def test_missing_library_fails_loudly(monkeypatch):
    from rawaudiovae_kelsey_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/librawvae_hip.so")
    with pytest.raises(_lib.RvError, match="no fallback"):
        _lib.lib()


def test_product_code_never_imports_the_oracle():
    bad = []
    for root in ("rawaudiovae_kelsey_amd", "rawvae"):
        for dp, _, files in os.walk(os.path.join(REPO, root)):
            for fn in files:
                if fn.endswith((".py", ".hip", ".h")):
                    with open(os.path.join(dp, fn)) as f:
                        txt = f.read()
                    if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "/root/reference" in txt.replace(
                            "(/root/reference/rawvae/model.py", ""):
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad


def test_header_is_plain_c_and_matches_the_library(tmp_path):
    """include/rawvae_hip.h must compile as C (no C++/HIP types leak into the ABI) and a C program
    linked against the library must resolve every entry point it declares."""
    import shutil
    import subprocess
    from rawaudiovae_kelsey_amd import _lib
    if not shutil.which("gcc"):
        pytest.skip("gcc not available")
    names = _header_symbols()
    src = tmp_path / "abi.c"
    body = "\n".join("  p[%d] = (void*)%s;" % (i, n) for i, n in enumerate(names))
    src.write_text('#include "rawvae_hip.h"\n#include <stdio.h>\nint main(void) {\n  void* p[%d];\n%s\n'
                   '  long bp, sp, hp, lp;\n  if (rv_pad_dims(4096, 1024, 2048, 64, &bp, &sp, &hp, &lp)) return 2;\n'
                   '  printf("%%d %%ld %%ld %%ld %%ld %%p\\n", rv_version(), bp, sp, hp, lp, p[0]);\n  return 0;\n}\n'
                   % (len(names), body))
    exe = tmp_path / "abi"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(REPO, "include"), str(src), "-o", str(exe),
                    "-L", libdir, "-l:librawvae_hip.so", "-Wl,-rpath," + libdir], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert out[:5] == ["100", "4096", "1024", "2048", "64"]


def test_host_logic_under_asan_ubsan(tmp_path):
    """SURVEY 5 (race detection / sanitizers): the native host code now exists, so its pure-host logic
    (pickers, descriptor tables, argument checks, error strings) runs under ASan + UBSan on the CPU.
    Device code is compiled normally (`-fno-gpu-sanitize`; GPU sanitizers are not available on this pool)."""
    import shutil
    import subprocess
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    src = os.path.join(REPO, "rawaudiovae_kelsey_amd", "csrc")
    flags = ["--offload-arch=gfx950", "-fsanitize=address,undefined", "-fno-gpu-sanitize", "-fno-omit-frame-pointer",
             "-g", "-O1", "-std=c++17", "-fPIC"]
    objs = []
    for name in ("gemm_launch", "elementwise", "plan", "linear_fp32", "latent"):
        o = str(tmp_path / (name + ".o"))
        r = subprocess.run(["hipcc"] + flags + ["-c", os.path.join(src, name + ".hip"), "-o", o],
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        objs.append(o)
    exe = str(tmp_path / "host_checks")
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-fsanitize=address,undefined", "-fno-gpu-sanitize", "-g",
                        "-I", os.path.join(REPO, "include"), "-x", "c++", os.path.join(REPO, "tests", "host", "host_checks.cpp"),
                        "-x", "none"] + objs + ["-o", exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="halt_on_error=1"))
    assert r.returncode == 0 and "0 failures" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]


def test_plan_descriptors_follow_the_heads_backward_in_use():
    """Host-only (rv_plan_create / bind / descs touch no device): with RV_OPT_LATENT_FUSED the heads' backward is the
    streaming kernel (one dWh slab and one fc1-bias partial row per 512 batch rows), without it the generic dual launch
    (the picker's split count, one partial row per dgrad row tile); the parameter descriptors the optimizer reads must
    switch with it, and no other tensor's may move."""
    import ctypes as C
    from rawaudiovae_kelsey_amd import _lib
    L = _lib.lib()
    for B in (4096, 1024):
        plan = C.c_void_p()
        L.rv_plan_create(C.byref(plan), B, 1024, 2048, 64)
        base = 0x10000000
        bufs = _lib.PlanBuffers(param=base, exp_avg=base + 0x4000000, exp_avg_sq=base + 0x8000000, grad=base + 0xc000000,
                                workspace=base + 0x10000000, step_counter=base + 0x100, loss_ring=base + 0x1000, ring=4)
        L.rv_plan_bind(plan, C.byref(bufs))
        arr = (_lib.ParamDesc * 10)()

        def splits():
            L.rv_plan_descs(plan, arr, 0)
            return [arr[i].grad_splits for i in range(10)]
        fused = splits()
        L.rv_plan_set_option(plan, _lib.OPT_LATENT_FUSED, 0)
        generic = splits()
        L.rv_plan_set_option(plan, _lib.OPT_LATENT_FUSED, 1)
        assert splits() == fused
        paired, bm, sp = _lib.dgrad_wgrad_pick(B, 2048, 128)
        assert fused[1] == fused[2] == fused[4] == B // 512
        assert generic[1] == B // bm and generic[2] == generic[4] == sp
        assert [v for i, v in enumerate(fused) if i not in (1, 2, 4)] == [v for i, v in enumerate(generic) if i not in (1, 2, 4)]
        L.rv_plan_destroy(plan)


def test_large_batch_plan_rules():
    """Host-only: the rules a large batch switches on (default.ini:27 trains at batch_size 131072).  The head biases'
    partial table keeps one row per 16 batch rows, but the optimizer's descriptors step over the rows the GEMM forms of
    the latent backward leave zero (one non-zero row per dz tile: 64 rows, 256 at large batches); fc3's weight gradient
    gets 16 K splits beside the 256 x 256 dz tiles; dgrad + wgrad are paired while their blocks come to one round of the
    256 CUs or two to four rounds filled to three quarters."""
    import ctypes as C
    from rawaudiovae_kelsey_amd import _lib
    L = _lib.lib()
    rows_expected = {(4096, 64): 16, (4096, 256): 64, (16384, 256): 64, (32768, 256): 256, (131072, 256): 256, (131072, 64): 64,
                     (131072, 128): 256}
    for (B, Lt), rows in rows_expected.items():
        plan = C.c_void_p()
        L.rv_plan_create(C.byref(plan), B, 1024, 2048, Lt)
        base = 0x10000000
        bufs = _lib.PlanBuffers(param=base, exp_avg=base + 0x4000000, exp_avg_sq=base + 0x8000000, grad=base + 0xc000000,
                                workspace=base + 0x10000000, step_counter=base + 0x100, loss_ring=base + 0x1000, ring=4)
        L.rv_plan_bind(plan, C.byref(bufs))
        arr = (_lib.ParamDesc * 10)()
        L.rv_plan_descs(plan, arr, 0)
        Lp = 64 if Lt <= 64 else 128 if Lt <= 128 else 256
        for i in (3, 5):
            assert arr[i].grad_splits == B // rows and arr[i].grad_split_stride == rows // 16 * 2 * Lp, (B, Lt, i)
        if (B, Lt) == (131072, 256):
            assert arr[6].grad_splits == 16          # dW3: 8 tiles of 256 x 256 x 16 K splits, first in the dz launch's grid
        L.rv_plan_set_option(plan, _lib.OPT_LATENT_FUSED, 0)     # the generic route: rv_reparam_bwd fills every row
        L.rv_plan_descs(plan, arr, 0)
        assert arr[3].grad_splits == arr[5].grad_splits == B // 16 and arr[3].grad_split_stride == 2 * Lp
        L.rv_plan_destroy(plan)
    # (batch, in features, out features) -> paired?, rounds of 256 blocks
    for (M, N, K), want in {(4096, 2048, 1024): (1, 1), (8192, 2048, 1024): (1, 2), (8192, 2048, 512): (1, 2),
                            (16384, 2048, 512): (1, 3), (16384, 2048, 1024): (1, 4), (32768, 2048, 1024): (0, 0),
                            (131072, 2048, 1024): (0, 0)}.items():
        paired, bm, sp = _lib.dgrad_wgrad_pick(M, N, K)
        assert paired == want[0], (M, N, K, paired, sp)
        if paired:
            blocks = (M // 256) * (N // 256) + (K // 256) * (N // 256) * sp
            assert -(-blocks // 256) == want[1] and bm == 256


def test_private_torch_entry_points_are_feature_tested_with_public_fallbacks(monkeypatch):
    """The drop-in loop's host path leans on three torch entry points outside the documented API
    (`torch.autograd.graph.increment_version`, `torch._foreach_add_`, `torch._C._cuda_getCurrentRawStream`).  Each is
    looked up once and has a public fall-back; their absence is simulated here (no GPU needed):
      * no raw-stream accessor -> `_lib.stream_ptr()` reads `torch.cuda.current_stream().cuda_stream`;
      * no `_foreach_add_` -> the ten step counters are bumped with `Tensor.add_`;
      * no `increment_version` -> the optimizer hook declines every step (PyTorch's own optimizer step runs: the public
        path) and says why."""
    import types

    from rawaudiovae_kelsey_amd import _lib, optim_hook
    # this torch has all three
    bump, add, missing = optim_hook._probe()
    assert bump is not None and missing == [] and optim_hook.missing == []
    # a torch with neither of the optimizer hook's two
    bare = types.SimpleNamespace(autograd=types.SimpleNamespace(graph=types.SimpleNamespace()))
    bump, add, missing = optim_hook._probe(bare)
    assert bump is None and missing == ["torch.autograd.graph.increment_version", "torch._foreach_add_"]
    counters = [torch.tensor(0.0), torch.tensor(4.0)]
    add(counters, 1)
    assert [float(c) for c in counters] == [1.0, 5.0]
    # the hook, with the version bump missing, leaves the step to PyTorch and records the reason
    monkeypatch.setattr(optim_hook, "_bump_versions", None)
    monkeypatch.setattr(optim_hook, "missing", ["torch.autograd.graph.increment_version"])
    monkeypatch.setattr(optim_hook, "enabled", True)
    monkeypatch.setitem(optim_hook._OWNER, -1, (lambda: None, lambda: None))     # "some model is registered"
    before = dict(optim_hook.stats["declined"])
    w = torch.nn.Parameter(torch.ones(3))
    opt = torch.optim.Adam([w], lr=0.5)
    w.grad = torch.ones(3)
    assert optim_hook._pre_step(opt, (opt,), {}) is None
    why = "this torch lacks torch.autograd.graph.increment_version"
    assert optim_hook.stats["declined"].get(why, 0) == before.get(why, 0) + 1
    opt.step()                                    # the stock step did the work
    assert torch.allclose(w.detach(), torch.full((3,), 0.5))
    # the stream accessor: private fast path when present, the public Stream object otherwise
    monkeypatch.setattr(_lib, "_raw_stream", None)
    monkeypatch.delattr(torch._C, "_cuda_getCurrentRawStream", raising=False)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: types.SimpleNamespace(cuda_stream=0x1234))
    assert _lib.stream_ptr() == 0x1234
    assert _lib.stream_ptr(types.SimpleNamespace(cuda_stream=0x77)) == 0x77
    monkeypatch.setattr(_lib, "_raw_stream", None)     # (the next caller looks the accessor up again)
