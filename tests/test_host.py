"""
CPU-only checks of the product's host side: the C-ABI library loads and exports
every symbol the header declares, the host-only entry points (integer tables,
AlphaBeta, rTA1 / rTA1L through a host-only handle) agree with the golden
vectors, and the Python host constants reproduce the reference's.
No GPU compute is called here.
"""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, golden
from oracle import sp_oracle as orc
from starry_process_amd import _lib, hostconst

LS = [5, 15, 20]


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "starry_process_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(sp_[A-Za-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    L = _lib.lib()
    for name in sorted(declared):
        assert hasattr(L, name), "libsp_hip.so does not export %s" % name
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    assert L.sp_version() >= 100
    assert L.sp_strerror(0) == b"ok"


def test_no_device_is_an_error_not_a_fallback():
    L = _lib.lib()
    if L.sp_device_count() > 0:
        pytest.skip("a GPU is visible")
    h = ctypes.c_void_p()
    assert L.sp_create(15, 2, 0, ctypes.byref(h)) == -3  # SP_ERR_NO_DEVICE
    # host-only handle: host entry points work, device entry points refuse
    _lib.check(L.sp_create(5, 2, -1, ctypes.byref(h)))
    x = np.zeros(8)
    assert L.sp_tensordotRz(h, _lib.hptr(x), _lib.hptr(x), 1, _lib.hptr(x), None) == -3
    assert L.sp_set_ylm_moments(h, _lib.hptr(x), _lib.hptr(x)) == -3
    # (round 6's entry points as well)
    assert L.sp_polar_moments_samples(h, 1, _lib.hptr(x), 1e-12, 1e-9, _lib.hptr(x), _lib.hptr(x), None) == -3
    assert L.sp_kernel_table_samples(h, 1, _lib.hptr(x), _lib.hptr(x), _lib.hptr(x), 1, 300, _lib.hptr(x), _lib.hptr(x),
                                     _lib.hptr(x), None) == -3
    assert L.sp_set_size_basis(h, _lib.hptr(x), _lib.hptr(x), 4, 300.0) == -3
    assert L.sp_plan_replicate(h, None, 2, None, ctypes.byref(ctypes.c_void_p())) == -3
    L.sp_destroy(h)
    with pytest.raises(_lib.SPError):
        from starry_process_amd.engine import Engine

        Engine(5, 2, 0)
    # ... and the gradient (grad.py: torch autograd around the library's reverse-mode kernels) has no CPU path either
    with pytest.raises(_lib.SPError):
        from starry_process_amd.grad import log_likelihood_with_grad

        log_likelihood_with_grad(np.zeros(256), np.eye(256), np.linspace(0, 1, 8), np.zeros(8), 1.0)


@pytest.mark.parametrize("L", LS)
def test_integer_tables_bit_exact(L):
    g = golden("ops_L%d" % L)
    tab = _lib.index_tables(L)
    ref = orc.index_tables(L)
    for k in ("l_of", "m_of", "mirror", "m0", "blk"):
        assert tab[k].dtype == np.int32 and np.array_equal(tab[k], ref[k])
    assert np.array_equal(tab["m_of"], g["tab_m_of"])
    assert np.array_equal(tab["mirror"], g["tab_mirror"])
    assert tab["blk"][-1] == int(g["nwig"])
    it, ir = _lib.wigner_int_tables(L), orc.wigner_int_tables(L)
    for k in it:
        assert np.array_equal(it[k], ir[k])


def test_alpha_beta_bit_exact():
    g = golden("norm")
    for z, v20, v10 in zip(g["z"], g["abN20"], g["abN10"]):
        assert np.array_equal(np.array(_lib.alpha_beta(z, 20)), v20)
        assert np.array_equal(np.array(_lib.alpha_beta(z, 10)), v10)


@pytest.mark.parametrize("L", LS)
def test_flux_operator_host(L):
    g = golden("ops_L%d" % L)
    lib = _lib.lib()
    h = ctypes.c_void_p()
    _lib.check(lib.sp_create(L, 2, -1, ctypes.byref(h)))
    N = (L + 1) ** 2
    out = np.empty(N)
    _lib.check(lib.sp_rTA1(h, _lib.hptr(out)))
    assert np.array_equal(out, g["rTA1"])
    us = np.ascontiguousarray(g["rTA1L_u"])
    o2 = np.empty((len(us), N))
    _lib.check(lib.sp_rTA1L(h, _lib.hptr(us), len(us), _lib.hptr(o2)))
    assert np.max(np.abs(o2 - g["rTA1L"])) < 1e-13
    # reverse mode (flux.h:529-557) against the reference's kernel and a finite difference
    gr = golden("rev_L%d" % L)
    rng = np.random.RandomState(int(gr["seed"]))
    K = int(gr["K"])
    # the generator's draw order (make_golden.py: gen_rev)
    rng.randn(K, N), rng.uniform(-7, 7, K), rng.randn(K, N), rng.randn(N, N), rng.randn(N, N), rng.randn(K)
    bfu = rng.randn(N)
    for u, bu_ref in zip(gr["ld_u"], gr["ld_bu"]):
        u = np.ascontiguousarray(u)
        bu = np.empty(2)
        _lib.check(lib.sp_rTA1L_rev(h, _lib.hptr(u), _lib.hptr(bfu), _lib.hptr(bu)))
        assert np.abs(bu - bu_ref).max() < (3e-10 if L <= 15 else 3e-8) * np.abs(bu_ref).max()
        for c in range(2):
            up, um = u.copy(), u.copy()
            up[c] += 1e-4   # the forward op carries up to 1e-10 of summation noise at L = 20
            um[c] -= 1e-4
            fp, fm = np.empty((1, N)), np.empty((1, N))
            _lib.check(lib.sp_rTA1L(h, _lib.hptr(up), 1, _lib.hptr(fp)))
            _lib.check(lib.sp_rTA1L(h, _lib.hptr(um), 1, _lib.hptr(fm)))
            fd = ((fp - fm)[0] * bfu).sum() / 2e-4
            assert abs(fd - bu[c]) < 1e-4 * max(1.0, abs(bu[c]))
    lib.sp_destroy(h)
    # udeg = 0 handle: rTA1L degenerates to rTA1 (flux.py:211-221)
    _lib.check(lib.sp_create(L, 0, -1, ctypes.byref(h)))
    o3 = np.empty((1, N))
    _lib.check(lib.sp_rTA1L(h, None, 1, _lib.hptr(o3)))
    assert np.array_equal(o3[0], g["rTA1"])
    lib.sp_destroy(h)


@pytest.mark.parametrize("L", LS)
def test_marginal_constants_match_reference(L):
    g = golden("consts_L%d" % L)
    assert np.array_equal(hostconst.G_matrix(L), g["G"])
    wnp, Wnp = hostconst.marginal_constants(L)
    gw = np.concatenate([g["wnp_%d" % l].reshape(-1) for l in range(L + 1)])
    assert np.max(np.abs(wnp - gw)) <= 1e-15 * np.max(np.abs(gw))
    assert np.max(np.abs(Wnp - g["Wnp"])) <= 1e-14 * np.max(np.abs(g["Wnp"]))
    dx, xp = hostconst.lag_grid(300)
    mom = golden("moments_L%d" % L)
    assert np.array_equal(xp, mom["default_u0_xp"]) and dx == float(mom["default_u0_dx"])


def test_bad_arguments_are_rejected():
    L = _lib.lib()
    h = ctypes.c_void_p()
    assert L.sp_create(0, 2, -1, ctypes.byref(h)) == -1
    assert L.sp_create(15, 9, -1, ctypes.byref(h)) == -1
    assert L.sp_index_tables(-1, None, None, None, None, None) == -1


def test_header_is_plain_c_and_links(tmp_path):
    """include/starry_process_amd.h is the drop-in boundary: it must compile as C (no C++, no
    torch types) and a C program must be able to link libsp_hip.so and call it.  Only host
    entry points are exercised (no GPU here)."""
    import shutil
    import subprocess

    gcc = shutil.which("gcc")
    if gcc is None or not os.path.exists(_lib.LIB_PATH):
        pytest.skip("needs gcc and the built library")
    src = tmp_path / "abi.c"
    src.write_text(r'''
#include <stdio.h>
#include "starry_process_amd.h"
int main(void) {
  int32_t l_of[36], m_of[36], mirror[36], m0[6], blk[7];
  if (sp_version() <= 0) return 1;
  if (sp_index_tables(5, l_of, m_of, mirror, m0, blk) != 0) return 2;
  if (l_of[35] != 5 || m_of[35] != 5 || m0[5] != 30) return 3;
  double al, be, dal, dbe;
  if (sp_alpha_beta(1e-3, 20, &al, &be, &dal, &dbe) != 0) return 4;
  sp_handle *h = 0;
  if (sp_create(5, 2, -1, &h) != 0) return 5;          /* host-only handle */
  double r[36];
  if (sp_rTA1(h, r) != 0) return 6;
  if (sp_lnlike_workspace_bytes(h, 1, 10, 1) <= 0) return 7;
  sp_star st = {1.0, 0.5, 0.0, 0.0, 0.0, 1e-6, 0, 0};
  if (sizeof(st) != 56) return 8;
  sp_destroy(h);
  printf("abi ok %d %.17g %.17g\n", sp_version(), al, r[0]);
  return 0;
}
''')
    exe = tmp_path / "abi"
    inc = os.path.join(ROOT, "include")
    libdir = os.path.dirname(_lib.LIB_PATH)
    cmd = [gcc, "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", inc, str(src), "-o", str(exe),
           "-L", libdir, "-l:libsp_hip.so", "-Wl,-rpath," + libdir, "-Wl,--allow-shlib-undefined"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    env = dict(os.environ)
    import torch

    tl = os.path.join(os.path.dirname(torch.__file__), "lib")
    env["LD_LIBRARY_PATH"] = tl + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True, env=env, timeout=120)
    assert out.stdout.startswith("abi ok")


def test_inline_dpp_instructions_keep_their_wait_states(tmp_path):
    """The v_fmac_f64_dpp chains of sp_diag.h are inline assembly, which the compiler's hazard recogniser does not look
    into: tools/check_dpp_hazard.py compiles the kernels that use them and finds no VALU write of a DPP source less
    than two wait states ahead of its read; the scan is checked on a hand-written listing of the case it was written for."""
    import importlib.util
    import shutil

    if not shutil.which("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    spec = importlib.util.spec_from_file_location("check_dpp_hazard", os.path.join(ROOT, "tools", "check_dpp_hazard.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    res = chk.check()
    for f, (seen, bad) in res.items():
        assert seen > 500, f
        assert bad == [], (f, bad[:3])
    # the scan itself, on a hand-written listing: the select directly in front of the DPP read (what the compiler did
    # once to a column of the pivot block's inverse), the same with idle states in between, and an unrelated register
    lst = tmp_path / "x.s"
    lst.write_text("\n".join([
        "v_cndmask_b32_e64 v10, 0, v18, s[4:5]",
        "v_fmac_f64_dpp v[10:11], v[10:11], -v[38:39] row_newbcast:12 row_mask:0xf bank_mask:0xf",
        "v_cndmask_b32_e64 v12, 0, v18, s[4:5]",
        "s_nop 1",
        "v_fmac_f64_dpp v[12:13], v[12:13], -v[38:39] row_newbcast:12 row_mask:0xf bank_mask:0xf",
        "v_mul_f64 v[20:21], v[2:3], v[4:5]",
        "v_fmac_f64_dpp v[14:15], v[14:15], -v[38:39] row_newbcast:12 row_mask:0xf bank_mask:0xf",
        "v_mul_f64 v[16:17], v[2:3], v[4:5]",
        "v_add_f64 v[30:31], v[2:3], v[4:5]",
        "v_fmac_f64_dpp v[16:17], v[16:17], -v[38:39] row_newbcast:12 row_mask:0xf bank_mask:0xf"]) + "\n")
    seen, bad = chk.scan(str(lst))
    assert seen == 4 and len(bad) == 2
    assert bad[0][0].startswith("v_cndmask_b32_e64 v10") and bad[1][0].startswith("v_mul_f64 v[16:17]")
