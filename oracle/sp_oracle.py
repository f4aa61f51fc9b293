"""
ORACLE -- TEST INFRASTRUCTURE ONLY.

CPU restatement (NumPy/SciPy + the plain-C ``oracle/sp_oracle.c``) of the
reference's ``StarryProcess.log_likelihood`` hot path: SURVEY.md section 8(a),
rows a1-a20.  It is the *checker* for the HIP product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it; ``starry_process_amd/`` never does.

Parity status: PINNED against the executed reference.  ``tests/golden/*.npz``
hold inputs/outputs produced by running the reference's own unmodified Python
(``/root/reference/starry_process/*.py``) on an eager Theano stand-in together
with the reference's own C++ compiled from its own headers
(``tests/golden/make_golden.py``); ``tests/test_oracle_golden.py`` checks every
function below against them.

Each function cites the reference file:line it follows.  Written for clarity,
with plain loops where the reference has loops; not optimised.
"""
import ctypes
import os
import subprocess

import numpy as np
import scipy.linalg
from scipy.special import gamma, hyp2f1

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBPATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def build():
    """Compile oracle/sp_oracle.c -> oracle/liboracle.so (gcc, seconds)."""
    subprocess.check_call(["make", "-C", _HERE, "liboracle.so"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIBPATH):
            build()
        _lib = ctypes.CDLL(_LIBPATH)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


# ---------------------------------------------------------------------------
# a1: integer layout tables
# ---------------------------------------------------------------------------
def nwig(l):
    """wigner.h:22-24 (also flux.py:77, ops/wigner/Rx.py:23-26)."""
    return ((l + 1) * (2 * l + 1) * (2 * l + 3)) // 3


def index_tables(ydeg):
    """l(n), m(n), mirror(n)=n(l,-m) (wigner.h:336), m0(l)=l^2+l (flux.py:200),
    packed-Wigner block offsets nwig(l-1) (wigner.h:22-30)."""
    N = (ydeg + 1) ** 2
    l_of = np.empty(N, dtype=np.int32)
    m_of = np.empty(N, dtype=np.int32)
    mirror = np.empty(N, dtype=np.int32)
    m0 = np.empty(ydeg + 1, dtype=np.int32)
    blk = np.empty(ydeg + 2, dtype=np.int32)
    lib().orc_index_tables(
        ydeg, _p(l_of), _p(m_of), _p(mirror), _p(m0), _p(blk)
    )
    return dict(l_of=l_of, m_of=m_of, mirror=mirror, m0=m0, blk=blk)


def wigner_int_tables(ydeg):
    """Integer cos/sin(k pi/2) factors of the x-rotation (wigner.h:232-270)."""
    out = [np.empty(ydeg + 1, dtype=np.int32) for _ in range(5)]
    lib().orc_wigner_int_tables(ydeg, *[_p(o) for o in out])
    return dict(zip(["cosmal", "sinmal", "sgn", "cosmga", "sinmga"], out))


# ---------------------------------------------------------------------------
# a2, a3, a9, a12: Wigner ops (C restatement)
# ---------------------------------------------------------------------------
def Rx(ydeg, theta):
    """RxOp(ydeg)(theta) -> (R, dR/dtheta), packed (wigner.h:145-284)."""
    R = np.empty(nwig(ydeg))
    dR = np.empty(nwig(ydeg))
    lib().orc_Rx(ydeg, ctypes.c_double(float(theta)), _p(R), _p(dR))
    return R, dR


def tensordotRz(ydeg, M, theta):
    """tensordotRzOp(ydeg)(M, theta) (wigner.h:289-339)."""
    M = _f64(M)
    theta = _f64(theta)
    K = theta.shape[0]
    N = (ydeg + 1) ** 2
    assert M.shape == (K, N)
    f = np.empty((K, N))
    lib().orc_tensordotRz(ydeg, _p(M), _p(theta), K, _p(f))
    return f


def special_tensordotRz(ydeg, T, M, theta):
    """special_tensordotRzOp(ydeg)(T, M, theta) (wigner.h:409-459)."""
    T = _f64(T)
    M = _f64(M)
    theta = _f64(theta)
    N = (ydeg + 1) ** 2
    assert T.shape == (N, N) and M.shape == (N, N)
    K = theta.shape[0]
    f = np.empty(K)
    lib().orc_special_tensordotRz(ydeg, _p(T), _p(M), _p(theta), K, _p(f))
    return f


def dotRx(ydeg, M, Rpacked):
    """FluxIntegral._dotRx: M . blockdiag(R^l) (flux.py:74-86)."""
    M = _f64(M)
    Rpacked = _f64(Rpacked)
    rows, N = M.shape
    assert N == (ydeg + 1) ** 2 and Rpacked.shape == (nwig(ydeg),)
    f = np.empty((rows, N))
    lib().orc_dotRx(ydeg, _p(M), rows, _p(Rpacked), _p(f))
    return f


# ---------------------------------------------------------------------------
# a5: flux operator rTA1 and rTA1L(u)  (flux.h)
# ---------------------------------------------------------------------------
def _rT(deg):
    """Phase-curve solution vector (flux.h:22-68)."""
    rT = np.zeros((deg + 1) ** 2)
    amp0 = np.pi
    lfac1 = 1.0
    lfac2 = 2.0 / 3.0
    for l in range(0, deg + 1, 4):
        amp = amp0
        for m in range(0, l + 1, 4):
            mu = l - m
            nu = l + m
            rT[l * l + l + m] = amp * lfac1
            rT[l * l + l - m] = amp * lfac1
            if l < deg:
                rT[(l + 1) * (l + 1) + l + m + 1] = amp * lfac2
                rT[(l + 1) * (l + 1) + l - m + 1] = amp * lfac2
            amp *= (nu + 2.0) / (mu - 2.0)
        lfac1 /= (l // 2 + 2) * (l // 2 + 3)
        lfac2 /= (l // 2 + 2.5) * (l // 2 + 3.5)
        amp0 *= 0.0625 * (l + 2) * (l + 2)
    amp0 = 0.5 * np.pi
    lfac1 = 0.5
    lfac2 = 4.0 / 15.0
    for l in range(2, deg + 1, 4):
        amp = amp0
        for m in range(2, l + 1, 4):
            mu = l - m
            nu = l + m
            rT[l * l + l + m] = amp * lfac1
            rT[l * l + l - m] = amp * lfac1
            if l < deg:
                rT[(l + 1) * (l + 1) + l + m + 1] = amp * lfac2
                rT[(l + 1) * (l + 1) + l - m + 1] = amp * lfac2
            amp *= (nu + 2.0) / (mu - 2.0)
        lfac1 /= (l // 2 + 2) * (l // 2 + 3)
        lfac2 /= (l // 2 + 2.5) * (l // 2 + 3.5)
        amp0 *= 0.0625 * l * (l + 4)
    return rT


def _polymulz(deg, p):
    """Multiply a polynomial (column vector in the (l,m) basis) by z
    (flux.h:74-96).  `p` has (deg+2)^2 rows; rows of degree <= deg are read."""
    pz = np.zeros_like(p)
    n = 0
    for l in range(deg + 1):
        for m in range(-l, l + 1):
            lz = l + 1
            nz = lz * lz + lz + m
            if (l + m) % 2 != 0:
                pz[nz - 4 * lz + 2] += p[n]
                pz[nz - 2] -= p[n]
                pz[nz + 2] -= p[n]
            else:
                pz[nz] += p[n]
            n += 1
    return pz


def _legendre_terms(deg):
    """P(z) part of each Ylm as {(l, m): value} lists (flux.h:102-152)."""
    N = (deg + 1) ** 2
    dns = np.zeros((N, N))
    term = 1.0
    fac = 1.0
    for m in range(deg + 1):
        dns[0, m * m + 2 * m] = fac
        dns[0, m * m] = fac
        # (flux.h:118-124 also seeds the l = m+1 column; the recursion below
        #  overwrites exactly that column, so the seed has no effect.)
        for l in range(m + 1, deg + 1):
            ip = l * l + l + m
            im = l * l + l - m
            colvec = _polymulz(deg - 1, dns[:, (l - 1) * (l - 1) + l - 1 + m])
            dns[:, ip] = (2 * l - 1) * colvec / (l - m)
            if l > m + 1:
                dns[:, ip] -= (
                    (l + m - 1) * dns[:, (l - 2) * (l - 2) + l - 2 + m] / (l - m)
                )
            dns[:, im] = dns[:, ip]
        fac *= -term
        term += 2
    out = []
    for col in range(N):
        terms = []
        n2 = 0
        for l in range(deg + 1):
            for m in range(-l, l + 1):
                if dns[n2, col] != 0:
                    terms.append((l, m, dns[n2, col]))
                n2 += 1
        out.append(terms)
    return out


def _theta_terms(deg):
    """theta(x, y) part of each Ylm (flux.h:158-183)."""
    N = (deg + 1) ** 2
    out = [[] for _ in range(N)]
    for m in range(deg + 1):
        term1 = 1.0
        term2 = float(m)
        for j in range(0, m + 1, 2):
            if j > 0:
                term1 *= -(m - j + 1.0) * (m - j + 2.0) / (j * (j - 1.0))
                term2 *= -(m - j) * (m - j + 1.0) / (j * (j + 1.0))
            for l in range(m, deg + 1):
                n1 = l * l + l + m
                n2 = l * l + l - m
                out[n1].append((m, 2 * j - m, term1))
                if j < m:
                    out[n2].append((m, 2 * (j + 1) - m, term2))
    return out


def _amp(deg):
    """Ylm amplitudes, one per column (flux.h:189-203)."""
    N = (deg + 1) ** 2
    a = np.zeros(N)
    for l in range(deg + 1):
        a[l * l + l] = np.sqrt(2 * (2 * l + 1))
        for m in range(1, l + 1):
            a[l * l + l + m] = -a[l * l + l + m - 1] / np.sqrt(
                (l + m) * (l - m + 1)
            )
            a[l * l + l - m] = a[l * l + l + m]
        a[l * l + l] *= np.sqrt(0.5)
    return a / (2 * np.sqrt(np.pi))


def _A1(deg):
    """Dense change-of-basis matrix Ylm -> polynomial (flux.h:206-279)."""
    N = (deg + 1) ** 2
    norm = 2.0 / np.sqrt(np.pi)
    C = _amp(deg)
    tZ = _legendre_terms(deg)
    tXY = _theta_terms(deg)
    A1 = np.zeros((N, N))
    for col in range(N):
        for (l1, m1, v1) in tZ[col]:
            odd1 = (l1 + m1) % 2 != 0
            for (l2, m2, v2) in tXY[col]:
                prod = v1 * v2
                if odd1 and ((l2 + m2) % 2 != 0):
                    trip = [
                        (l1 + l2 - 2, m1 + m2, prod),
                        (l1 + l2, m1 + m2 - 2, -prod),
                        (l1 + l2, m1 + m2 + 2, -prod),
                    ]
                else:
                    trip = [(l1 + l2, m1 + m2, prod)]
                for (l, m, v) in trip:
                    A1[l * l + l + m, col] += v * norm * C[col]
    return A1


_CONST = {}


def rTA1(ydeg):
    """rTA1Op(ydeg)() (flux.h:302-309)."""
    key = ("rTA1", ydeg)
    if key not in _CONST:
        _CONST[key] = _rT(ydeg) @ _A1(ydeg)
    return _CONST[key].copy()


def _limbdark_setup(ydeg, udeg):
    """LimbDark constructor: rT, A1 at degree ydeg+udeg, U1 (flux.h:332-409,
    483-494)."""
    key = ("LD", ydeg, udeg)
    if key in _CONST:
        return _CONST[key]
    LU = ydeg + udeg
    rT = _rT(LU)
    A1 = _A1(LU)
    norm = 2.0 / np.sqrt(np.pi)
    LT = np.zeros((LU + 1, LU + 1))
    YT = np.zeros((LU + 1, LU + 1))
    for l in range(LU + 1):
        lchoosek = 1.0
        for k in range(l + 1):
            LT[k, l] = lchoosek if (k + 1) % 2 == 0 else -lchoosek
            lchoosek *= (l - k) / (k + 1.0)
    twol, lfac, fac0 = 1.0, 1.0, 1.0
    for l in range(0, LU + 1, 2):
        amp = twol * np.sqrt((2 * l + 1) / (4 * np.pi)) / lfac
        lchoosek = 1.0
        fac = fac0
        for k in range(0, l + 1, 2):
            YT[k, l] = amp * lchoosek * fac
            fac *= (k + l + 1.0) / (k - l + 1.0)
            lchoosek *= (l - k) * (l - k - 1) / ((k + 1.0) * (k + 2.0))
        fac0 *= -0.25 * (l + 1) * (l + 1)
        lfac *= (l + 1.0) * (l + 2.0)
        twol *= 4.0
    twol, lfac, fac0 = 2.0, 1.0, 0.5
    for l in range(1, LU + 1, 2):
        amp = twol * np.sqrt((2 * l + 1) / (4 * np.pi)) / lfac
        lchoosek = float(l)
        fac = fac0
        for k in range(1, l + 1, 2):
            YT[k, l] = amp * lchoosek * fac
            fac *= (k + l + 1.0) / (k - l + 1.0)
            lchoosek *= (l - k) * (l - k - 1) / ((k + 1.0) * (k + 2.0))
        fac0 *= -0.25 * (l + 2) * l
        lfac *= (l + 1.0) * (l + 2.0)
        twol *= 4.0
    U0 = np.linalg.solve(YT, LT) / norm
    NLU = (LU + 1) ** 2
    X = np.zeros((NLU, LU + 1))
    for l in range(LU + 1):
        X[l * (l + 1), l] = 1
    U1 = (A1 @ (X @ U0))[: (udeg + 1) ** 2, : udeg + 1]
    N = (ydeg + 1) ** 2
    _CONST[key] = (rT, A1[:N, :N].copy(), U1)
    return _CONST[key]


def rTA1L(ydeg, udeg, u):
    """rTA1LOp(ydeg, udeg)(u) (flux.h:500-523, 415-441)."""
    rT, A1NbyN, U1 = _limbdark_setup(ydeg, udeg)
    u_ = np.concatenate(([-1.0], np.asarray(u, dtype=float)[:udeg]))
    p = U1 @ u_
    norm = 1.0 / np.dot(rT[: (udeg + 1) ** 2], p)
    p = p * (norm * np.pi)
    N = (ydeg + 1) ** 2
    NLU = (ydeg + udeg + 1) ** 2
    Lp = np.zeros((NLU, N))
    n1 = 0
    for l1 in range(ydeg + 1):
        for m1 in range(-l1, l1 + 1):
            odd1 = (l1 + m1) % 2 != 0
            n2 = 0
            for l2 in range(udeg + 1):
                for m2 in range(-l2, l2 + 1):
                    l = l1 + l2
                    n = l * l + l + m1 + m2
                    if odd1 and ((l2 + m2) % 2 != 0):
                        Lp[n - 4 * l + 2, n1] += p[n2]
                        Lp[n - 2, n1] -= p[n2]
                        Lp[n + 2, n1] -= p[n2]
                    else:
                        Lp[n, n1] += p[n2]
                    n2 += 1
            n1 += 1
    return (rT @ Lp) @ A1NbyN


# ---------------------------------------------------------------------------
# a6: inclination-marginalisation constants (flux.py:107-179, wigner.py)
# ---------------------------------------------------------------------------
def _polyprod(x1, x2):
    """wigner.py:166-174."""
    out = np.zeros(len(x1) + len(x2) - 1)
    for i, a in enumerate(x1):
        for j, b in enumerate(x2):
            out[i + j] += a * b
    return out


def wigner_R_poly(ydeg, c1=0, s1=1, c3=0, s3=-1):
    """Real Wigner matrices whose entries are coefficient vectors in the basis
    sin(phi/2)^(2l-i) cos(phi/2)^i, i = 0..2l (wigner.py:192-372; the default
    Euler angles are those flux.py:49-51 passes)."""
    r2 = np.sqrt(2.0)
    D = [np.full((2 * l + 1,) * 3, np.nan) for l in range(ydeg + 1)]
    R = [np.full((2 * l + 1,) * 3, np.nan) for l in range(ydeg + 1)]
    D[0][0, 0] = [1]
    R[0][0, 0] = [1]
    if ydeg == 0:
        return R
    D[1][2, 2] = [0, 0, 1]
    D[1][2, 1] = [0, -r2, 0]
    D[1][2, 0] = [1, 0, 0]
    D[1][1, 2] = -D[1][2, 1]
    D[1][1, 1] = D[1][2, 2] - D[1][2, 0]
    D[1][1, 0] = D[1][2, 1]
    D[1][0, 2] = D[1][2, 0]
    D[1][0, 1] = D[1][1, 2]
    D[1][0, 0] = D[1][2, 2]
    cosag = c1 * c3 - s1 * s3
    cosamg = c1 * c3 + s1 * s3
    sinag = s1 * c3 + c1 * s3
    sinamg = s1 * c3 - c1 * s3
    R[1][1, 1] = D[1][1, 1]
    R[1][2, 1] = r2 * D[1][1, 2] * c1
    R[1][0, 1] = r2 * D[1][1, 2] * s1
    R[1][1, 2] = r2 * D[1][2, 1] * c3
    R[1][1, 0] = -r2 * D[1][2, 1] * s3
    R[1][2, 2] = D[1][2, 2] * cosag - D[1][2, 0] * cosamg
    R[1][2, 0] = -D[1][2, 2] * sinag - D[1][2, 0] * sinamg
    R[1][0, 2] = D[1][2, 2] * sinag - D[1][2, 0] * sinamg
    R[1][0, 0] = D[1][2, 2] * cosag + D[1][2, 0] * cosamg

    for l in range(2, ydeg + 1):
        lo, hi = 1 - l, l - 1
        # last row (wigner.py:197-206)
        D[l][2 * l, 2 * l] = _polyprod(D[l - 1][2 * l - 2, 2 * l - 2], [0, 0, 1])
        D[l][2 * l, 0] = _polyprod(D[l - 1][2 * l - 2, 0], [1, 0, 0])
        for m in range(hi, lo - 1, -1):
            v = -np.sqrt((l + m + 1.0) / (l - m)) * D[l][2 * l, m + 1 + l]
            D[l][2 * l, m + l] = np.append(v[1:], [0])
        # upper quarter triangle (wigner.py:210-236)
        for mp in range(l - 1, -1, -1):
            laux, lbux = l + mp, l - mp
            aux = 1.0 / ((l - 1) * np.sqrt(laux * lbux))
            cux = np.sqrt((laux - 1) * (lbux - 1)) * l
            for m in range(hi, lo - 1, -1):
                lauz, lbuz = l + m, l - m
                auz = 1.0 / np.sqrt(lauz * lbuz)
                fact = aux * auz
                a = l * (l - 1)
                b = -(m * mp) / a
                D[l][mp + l, m + l] = _polyprod(
                    fact * (2 * l - 1) * a * D[l - 1][mp + l - 1, m + l - 1],
                    [b - 1, 0, b + 1],
                )
                if lbuz != 1 and lbux != 1:
                    cuz = np.sqrt((lauz - 1) * (lbuz - 1))
                    D[l][mp + l, m + l] -= (fact * cux * cuz) * _polyprod(
                        D[l - 2][mp + l - 2, m + l - 2], [1, 0, 2, 0, 1]
                    )
            lo += 1
            hi -= 1
        # reflection / inversion (wigner.py:243-263)
        sign = 1
        lo, hi = -l, l - 1
        for m in range(l, 0, -1):
            for mp in range(lo, hi + 1):
                D[l][mp + l, m + l] = sign * D[l][m + l, mp + l]
                sign *= -1
            lo += 1
            hi -= 1
        lo = -l
        hi = lo
        for m in range(l - 1, -(l + 1), -1):
            sign = -1
            for mp in range(hi, lo - 1, -1):
                D[l][mp + l, m + l] = sign * D[l][-mp + l, -m + l]
                sign *= -1
            hi += 1
        # complex -> real (wigner.py:265-292)
        R[l][l, l] = D[l][l, l]
        cosmal, sinmal, sign = c1, s1, -1
        for mp in range(1, l + 1):
            cosmga, sinmga = c3, s3
            aux = r2 * D[l][l, mp + l]
            R[l][mp + l, l] = aux * cosmal
            R[l][-mp + l, l] = aux * sinmal
            for m in range(1, l + 1):
                aux = r2 * D[l][m + l, l]
                R[l][l, m + l] = aux * cosmga
                R[l][l, -m + l] = -aux * sinmga
                d1 = D[l][-mp + l, -m + l]
                d2 = sign * D[l][mp + l, -m + l]
                cag = cosmal * cosmga - sinmal * sinmga
                cagm = cosmal * cosmga + sinmal * sinmga
                sag = sinmal * cosmga + cosmal * sinmga
                sagm = sinmal * cosmga - cosmal * sinmga
                R[l][mp + l, m + l] = d1 * cag + d2 * cagm
                R[l][mp + l, -m + l] = -d1 * sag + d2 * sagm
                R[l][-mp + l, m + l] = d1 * sag + d2 * sagm
                R[l][-mp + l, -m + l] = d1 * cag - d2 * cagm
                aux = cosmga * c3 - sinmga * s3
                sinmga = sinmga * c3 + cosmga * s3
                cosmga = aux
            sign *= -1
            aux = cosmal * c1 - sinmal * s1
            sinmal = sinmal * c1 + cosmal * s1
            cosmal = aux
    return R


def G_matrix(ydeg):
    """G[j, i] = int_0^{pi/2} cos(x/2)^i' sin(x/2)^j' sin x dx as the reference
    tabulates it: G = [[_G(i, j) for i] for j] (flux.py:107-137)."""
    n = 4 * ydeg + 1

    def _G(j, i):
        return 2 * gamma(1 + 0.5 * i) * gamma(1 + 0.5 * j) / gamma(
            0.5 * (4 + i + j)
        ) - (2 ** (1 - 0.5 * i) / (2 + i)) * hyp2f1(
            1 + 0.5 * i, -0.5 * j, 2 + 0.5 * i, 0.5
        )

    return np.array([[_G(i, j) for i in range(n)] for j in range(n)])


def precompute(ydeg):
    """wnp[l] ((2l+1) vectors... see below), Wnp (N x N)  (flux.py:121-179).

    Returns (G, wnp, Wnp) with wnp a list of (2l+1, 2l+1) arrays."""
    key = ("pre", ydeg)
    if key in _CONST:
        return _CONST[key]
    N = (ydeg + 1) ** 2
    Rp = wigner_R_poly(ydeg)
    G = G_matrix(ydeg)
    wnp = []
    for l in range(ydeg + 1):
        m = np.arange(-l, l + 1)
        wnp.append(Rp[l] @ G[l - m, l + m])
    Q = np.empty((2 * ydeg + 1, 2 * ydeg + 1, 2 * ydeg + 1, N))
    for l1 in range(ydeg + 1):
        k = np.arange(l1 ** 2, (l1 + 1) ** 2)
        k0 = np.arange(2 * l1 + 1).reshape(-1, 1)
        for p in range(N):
            l2 = int(np.floor(np.sqrt(p)))
            j = np.arange(l2 ** 2, (l2 + 1) ** 2)
            j0 = np.arange(2 * l2 + 1).reshape(1, -1)
            L = Rp[l1][l1, k - l1 ** 2] @ G[k0 + j0, 2 * l1 - k0 + 2 * l2 - j0]
            Rr = Rp[l2][j - l2 ** 2, p - l2 ** 2].T
            Q[l1, : 2 * l1 + 1, : 2 * l2 + 1, p] = L @ Rr
    Wnp = np.empty((N, N))
    for l1 in range(ydeg + 1):
        i = np.arange(l1 ** 2, (l1 + 1) ** 2)
        for l2 in range(ydeg + 1):
            j = np.arange(l2 ** 2, (l2 + 1) ** 2)
            Wnp[i.reshape(-1, 1), j.reshape(1, -1)] = Q[
                l1, : 2 * l1 + 1, l2, j
            ].T
    _CONST[key] = (G, wnp, Wnp)
    return _CONST[key]


# ---------------------------------------------------------------------------
# a4, a7, a8: moments in the polar frame, inclination integrals
# ---------------------------------------------------------------------------
def polar_moments(ydeg, mean_ylm, cov_ylm):
    """ez = R^T mu, Ez = R^T (Sigma + mu mu^T) R, R = Rx(pi/2)
    (flux.py:54-62)."""
    Rpk = Rx(ydeg, 0.5 * np.pi)[0]
    mean_ylm = _f64(mean_ylm)
    ez = dotRx(ydeg, mean_ylm.reshape(1, -1), Rpk).T
    mom2 = np.ascontiguousarray(cov_ylm + np.outer(mean_ylm, mean_ylm))
    tmp = np.ascontiguousarray(dotRx(ydeg, mom2, Rpk).T)
    Ez = dotRx(ydeg, tmp, Rpk)
    return ez, Ez


def inclination_integrals(ydeg, rta1):
    """w[l] = rTA1[l-block] . wnp[l];  W = Wnp * outer(rTA1[m0], rTA1[m0])
    blockwise (flux.py:181-231)."""
    _, wnp, Wnp = precompute(ydeg)
    w = [rta1[l * l : (l + 1) ** 2] @ wnp[l] for l in range(ydeg + 1)]
    m0 = np.array([l * l + l for l in range(ydeg + 1)])
    Z = np.outer(rta1[m0], rta1[m0])
    W = np.zeros_like(Wnp)
    for l1 in range(ydeg + 1):
        i = np.arange(l1 ** 2, (l1 + 1) ** 2).reshape(-1, 1)
        for l2 in range(ydeg + 1):
            j = np.arange(l2 ** 2, (l2 + 1) ** 2).reshape(1, -1)
            W[i, j] = Wnp[i, j] * Z[l1, l2]
    return w, W


def marginal_mean_var(ydeg, w, W, ez, Ez):
    """flux.py:297-308."""
    mean = np.sum(
        [np.dot(w[l], ez[l * l : (l + 1) ** 2]) for l in range(ydeg + 1)]
    )
    var = np.tensordot(W, Ez) - mean ** 2
    return float(mean), float(var)


# ---------------------------------------------------------------------------
# a9, a10: kernel table + spline
# ---------------------------------------------------------------------------
def lag_grid(covpts):
    """dx, xp (flux.py:311-314)."""
    dx = 2 * np.pi / covpts
    xp = np.arange(-dx, 2 * np.pi + 2.5 * dx, dx)
    return dx, xp


def kernel_table(ydeg, W, Ez, mean, covpts):
    """yp and the cubic coefficients a0..a3 (flux.py:310-330)."""
    dx, xp = lag_grid(covpts)
    mom2 = special_tensordotRz(ydeg, W, Ez, xp)
    yp = mom2 - mean ** 2
    y0, y1, y2, y3 = yp[:-3], yp[1:-2], yp[2:-1], yp[3:]
    a0 = y1
    a1 = -y0 / 3.0 - 0.5 * y1 + y2 - y3 / 6.0
    a2 = 0.5 * (y0 + y2) - y1
    a3 = 0.5 * ((y1 - y2) + (y3 - y0) / 3.0)
    return dict(dx=dx, xp=xp, yp=yp, a0=a0, a1=a1, a2=a2, a3=a3)


# ---------------------------------------------------------------------------
# a11: K x K interpolation
# ---------------------------------------------------------------------------
def phase(t, p):
    """theta = 2 pi mod(t / p, 1) (flux.py:262, 279)."""
    return 2 * np.pi * np.mod(np.asarray(t, dtype=float).reshape(-1) / p, 1.0)


def interpolate_indices(t, p, dx):
    """The int64 segment index of every (i, j) pair (flux.py:262-264)."""
    theta = phase(t, p)
    x = np.abs(theta[:, None] - theta[None, :]).reshape(-1)
    return np.floor(x / dx).astype("int64")


def interpolate_cov(t, p, tab, var):
    """flux.py:256-276."""
    theta = phase(t, p)
    K = theta.shape[0]
    if K == 1:
        return np.array([[var]])
    dx, xp = tab["dx"], tab["xp"]
    x = np.abs(theta[:, None] - theta[None, :]).reshape(-1)
    inds = np.floor(x / dx).astype("int64")
    x0 = (x - xp[inds + 1]) / dx
    cov = (
        tab["a0"][inds]
        + tab["a1"][inds] * x0
        + tab["a2"][inds] * x0 ** 2
        + tab["a3"][inds] * x0 ** 3
    )
    return cov.reshape(K, K)


# ---------------------------------------------------------------------------
# a12, a13: conditional branch
# ---------------------------------------------------------------------------
def design_matrix(ydeg, rta1, t, inc_rad, p):
    """A = ((1_K x rTA1) . Rx(-i)) Rz(theta) Rx(pi/2) (flux.py:278-281,88-105).
    `inc_rad` in radians."""
    theta = phase(t, p)
    M = np.tile(rta1, (theta.shape[0], 1))
    M = dotRx(ydeg, M, Rx(ydeg, -inc_rad)[0])
    M = tensordotRz(ydeg, M, theta)
    M = dotRx(ydeg, M, Rx(ydeg, 0.5 * np.pi)[0])
    return M


def conditional_mean_cov(A, mean_ylm, cov_ylm):
    """flux.py:337-343."""
    mean = np.dot(A, mean_ylm)[0]
    cov = np.dot(np.dot(A, cov_ylm), A.T)
    return float(mean), cov


# ---------------------------------------------------------------------------
# a14: temporal kernels (temporal.py:8-16)
# ---------------------------------------------------------------------------
def ExpSquaredKernel(t1, t2, tau):
    dt = np.abs(np.reshape(t1, (-1, 1)) - np.reshape(t2, (1, -1)))
    return np.exp(-(dt ** 2) / (2 * tau))


def Matern32Kernel(t1, t2, tau):
    dt = np.abs(np.reshape(t1, (-1, 1)) - np.reshape(t2, (1, -1)))
    x = np.sqrt(3) * dt / tau
    return (1 + x) * np.exp(-x)


# ---------------------------------------------------------------------------
# a15: normalisation (ops/norm/norm.py:26-44, sp.py:705-727)
# ---------------------------------------------------------------------------
def alpha_beta(z, N=20):
    fac = 1.0
    alpha = 0.0
    beta = 0.0
    dadz = 0.0
    dbdz = 0.0
    dfdz = 0.0
    for n in range(0, N + 1):
        dadz += dfdz
        dbdz += 2 * n * dfdz
        dfdz = (2 * n + 3) * (dfdz * z + fac)
        alpha += fac
        beta += 2 * n * fac
        fac *= z * (2 * n + 3)
    return alpha, beta, dadz, dbdz


def normalize(mu, Sig, N=20):
    """Returns (normalised covariance, z)."""
    K = Sig.shape[0]
    j = np.ones((K, 1))
    m = np.mean(Sig)
    q = np.dot(Sig, j) / (K * m)
    z = m / mu ** 2
    p = j - q
    alpha, beta, _, _ = alpha_beta(z, N)
    ppT = np.dot(p, p.T)
    qqT = np.dot(q, q.T)
    normSig = (alpha / mu ** 2) * Sig + z * ((alpha + beta) * ppT - alpha * qqT)
    return normSig, float(z)


# ---------------------------------------------------------------------------
# a17, a18: Cholesky / solves with the reference's NaN semantics
# (math.py:20-38, 75-100)
# ---------------------------------------------------------------------------
def cho_factor(A):
    try:
        return scipy.linalg.cholesky(A, lower=True)
    except (scipy.linalg.LinAlgError, ValueError):
        return np.zeros(A.shape) * np.nan


def _solve_tri(A, b, lower):
    if np.any(np.isnan(A)) or np.any(np.isnan(b)):
        return np.ones_like(b) * np.nan
    return scipy.linalg.solve_triangular(A, b, lower=lower)


def cho_solve(L, b):
    return _solve_tri(L.T, _solve_tri(L, b, True), False)


# reverse mode of the two
def solve_L_op(A, b, c, c_bar, lower):
    """(A_bar, b_bar) of c = A^-1 b for triangular A (Solve.L_op, math.py:40-72:
    b_bar = A^-T c_bar, A_bar = -tri(b_bar c^T))."""
    b_bar = _solve_tri(A.T, c_bar, not lower)
    A_bar = -np.outer(b_bar, c) if c.ndim == 1 else -b_bar.dot(c.T)
    return (np.tril(A_bar) if lower else np.triu(A_bar)), b_bar


def cholesky_L_op(L, L_bar):
    """C_bar of L = cholesky(C), lower.  The reference inherits this from the Theano /
    Aesara ``slinalg.Cholesky`` it subclasses (math.py:75; theano-pymc 1.1.2 / aesara 2.x,
    not vendored, absent here); restated from its published formula (I. Murray,
    "Differentiation of the Cholesky decomposition", arXiv:1602.07527, eq. for the
    blocked-free reverse mode):
        Phi = tril(L^T L_bar) with the diagonal halved,  S = L^-T Phi L^-1,
        C_bar = tril(S + S^T) - diag(S);  NaN when the factor is NaN (on_error="nan").
    Pinned by finite differences in tests/test_linalg_rev.py."""
    if np.any(np.isnan(L)):
        return np.full(L.shape, np.nan)
    P = np.tril(L.T.dot(L_bar))
    P[np.diag_indices_from(P)] *= 0.5
    X = scipy.linalg.solve_triangular(L.T, P.T, lower=False)      # L^-T Phi^T
    S = scipy.linalg.solve_triangular(L.T, X.T, lower=False)      # L^-T Phi L^-1
    return np.tril(S + S.T) - np.diag(np.diag(S))


# ---------------------------------------------------------------------------
# a16, a19, a20: the whole evaluation
# ---------------------------------------------------------------------------
def check_bounds(name, value, lower, upper, tol=1e-6):
    """ops/exceptions.py:30-48."""
    v = np.asarray(value, dtype=float)
    if np.any((v < lower - tol) | (v > upper + tol)):
        raise ValueError("%s out of bounds" % name)
    return value


class OracleProcess(object):
    """The slice of StarryProcess that log_likelihood needs, fed with the
    Ylm moments (mu_y, Sigma_y) that the (out-of-scope) upstream integrals
    produce (sp.py:257-281, flux.py:23-72)."""

    def __init__(
        self,
        mean_ylm,
        cov_ylm,
        ydeg=15,
        udeg=2,
        marginalize_over_inclination=True,
        normalized=True,
        covpts=300,
        tau=None,
        temporal_kernel=Matern32Kernel,
        normalization_order=20,
        normalization_zmax=0.023,
    ):
        self.ydeg, self.udeg = ydeg, udeg
        self.N = (ydeg + 1) ** 2
        self.mean_ylm = _f64(mean_ylm)
        self.cov_ylm = _f64(cov_ylm)
        self.marg = marginalize_over_inclination
        self.normalized = normalized
        self.covpts = covpts
        self.tau = tau
        self.temporal_kernel = temporal_kernel
        self.normN = normalization_order
        self.zmax = normalization_zmax
        self.ez, self.Ez = polar_moments(ydeg, self.mean_ylm, self.cov_ylm)
        self.z = None

    def _rta1(self, u):
        if self.udeg > 0:
            return rTA1L(self.ydeg, self.udeg, np.asarray(u, float)[: self.udeg])
        return rTA1(self.ydeg)

    def flux_mean_cov(self, t, i=60.0, p=1.0, u=(0.0, 0.0)):
        """FluxIntegral._compute (flux.py:283-343); i in degrees."""
        t = np.asarray(t, dtype=float).reshape(-1)
        inc = check_bounds("i", i * np.pi / 180, 0, 0.5 * np.pi)
        check_bounds("p", p, 0, np.inf)
        rta1 = self._rta1(u)
        if self.marg:
            w, W = inclination_integrals(self.ydeg, rta1)
            mean, var = marginal_mean_var(self.ydeg, w, W, self.ez, self.Ez)
            tab = kernel_table(self.ydeg, W, self.Ez, mean, self.covpts)
            cov = interpolate_cov(t, p, tab, var)
            self.tab = tab
            self.var = var
        else:
            A = design_matrix(self.ydeg, rta1, t, inc, p)
            mean, cov = conditional_mean_cov(A, self.mean_ylm, self.cov_ylm)
        return mean, cov

    def mean(self, t, i=60.0, p=1.0, u=(0.0, 0.0)):
        t = np.asarray(t, dtype=float).reshape(-1)
        if self.normalized:
            return np.zeros_like(t)
        return self.flux_mean_cov(t, i, p, u)[0] * np.ones_like(t)

    def cov(self, t, i=60.0, p=1.0, u=(0.0, 0.0)):
        """sp.py:674-703."""
        t = np.asarray(t, dtype=float).reshape(-1)
        mean, cov = self.flux_mean_cov(t, i, p, u)
        if self.tau is not None:
            cov = cov * self.temporal_kernel(t, t, self.tau)
        if self.normalized:
            cov, self.z = normalize(1.0 + mean, cov, self.normN)
        return cov

    def log_likelihood(
        self,
        t,
        flux,
        data_cov,
        i=60.0,
        p=1.0,
        u=(0.0, 0.0),
        baseline_mean=0.0,
        baseline_var=0.0,
    ):
        """sp.py:1129-1188."""
        t = np.asarray(t, dtype=float).reshape(-1)
        gp_mean = self.mean(t, i, p, u)
        gp_cov = self.cov(t, i, p, u)
        K = gp_mean.shape[0]
        data_cov = np.asarray(data_cov, dtype=float)
        if data_cov.ndim == 0:
            C = data_cov * np.eye(K)
        elif data_cov.ndim == 1:
            C = np.diag(data_cov)
        else:
            C = data_cov
        gp_cov = gp_cov + C
        gp_cov = gp_cov + baseline_var
        L = cho_factor(gp_cov)
        mean = np.reshape(gp_mean + baseline_mean, (K, 1))
        r = np.reshape(np.transpose(np.asarray(flux, dtype=float)), (K, -1)) - mean
        M = r.shape[1]
        x = cho_solve(L, r)
        lnlike = -0.5 * np.sum(r * x)
        with np.errstate(invalid="ignore", divide="ignore"):
            lnlike -= M * np.sum(np.log(np.diag(L)))
        lnlike -= 0.5 * K * M * np.log(2 * np.pi)
        if self.normalized and self.z > self.zmax:
            lnlike = -np.inf
        if np.isnan(lnlike):
            lnlike = -np.inf
        return float(lnlike)


# ---------------------------------------------------------------------------
# Upstream moments by exact quadrature of rotations: CPU counterpart of
# starry_process_amd/upstream_device.py (same nodes, same sequence of rotations, the C
# restatements of Rx / dotRx / tensordotRz above).  It is an independent check of what
# the reference's integrals (latitude.py:199-212, longitude.py:19-24, contrast.py:18-33)
# ARE, and the checker of the device version.
def gauss_jacobi(n, a, b):
    """n-point Gauss-Jacobi rule for (1 - t)^a (1 + t)^b, weights summing to 1, by Golub-Welsch
    with LAPACK's tridiagonal eigensolver (scipy.special.roots_jacobi overflows in its
    normalisation for a + b > ~1 020; the normalisation cancels here)."""
    from scipy.linalg import eigh_tridiagonal

    k = np.arange(n, dtype=np.float64)
    s = 2.0 * k + a + b
    d = np.empty(n)
    d[0] = (b - a) / (a + b + 2.0)
    d[1:] = (b * b - a * a) / (s[1:] * (s[1:] + 2.0))
    k1, s1 = k[1:], s[1:]
    with np.errstate(invalid="ignore", divide="ignore"):
        e2 = 4.0 * k1 * (k1 + a) * (k1 + b) * (k1 + a + b) / (s1 * s1 * (s1 + 1.0) * (s1 - 1.0))
    if n > 1:
        e2[0] = 4.0 * (1.0 + a) * (1.0 + b) / ((2.0 + a + b) ** 2 * (3.0 + a + b))
    if n == 1:
        return d.copy(), np.ones(1)
    t, V = eigh_tridiagonal(d, np.sqrt(e2))
    w = V[0] ** 2
    return t, w / w.sum()


def ylm_moments_quadrature(size_first, size_factor_cols, alpha, beta, c, n, ydeg,
                           epsy=1e-12, epsy15=1e-9, refine=1):
    """size_first [N]; size_factor_cols [m, N] (columns of the size second-moment factor;
    pass size_first[None, :] for a fixed spot radius).  Returns (mu_y, Sigma_y)."""
    N = (ydeg + 1) ** 2
    nq = refine * (ydeg + 2)
    t, w = gauss_jacobi(nq, beta - 1.0, alpha - 1.0)
    x = 0.5 * (1.0 + t)
    phis = np.concatenate([np.arccos(x), -np.arccos(x)])
    wphi = 0.5 * np.concatenate([w, w])
    nl = refine * (2 * ydeg + 3)
    lams = 2.0 * np.pi * np.arange(nl) / nl
    Rp, Rm = Rx(ydeg, 0.5 * np.pi)[0], Rx(ydeg, -0.5 * np.pi)[0]
    vecs = np.vstack([_f64(size_first)[None, :], _f64(size_factor_cols)])
    mom1 = np.zeros(N)
    mom2 = np.zeros((N, N))
    for ph, wk in zip(phis, wphi):
        V = dotRx(ydeg, vecs, Rx(ydeg, ph)[0])
        U = dotRx(ydeg, V, Rp)
        for j in range(vecs.shape[0]):
            A = dotRx(ydeg, tensordotRz(ydeg, np.repeat(U[j:j + 1], nl, axis=0), lams), Rm)
            if j == 0:
                mom1 += (wk / nl) * A.sum(axis=0)
            else:
                mom2 += (wk / nl) * (A.T @ A)
    mean = np.pi * c * n * mom1
    cov = (np.pi * c) ** 2 * n * (mom2 - np.outer(mom1, mom1))
    lam = np.ones(N) * epsy
    lam[15 ** 2:] = epsy15
    return mean, cov + np.diag(lam)
