"""The run-time-length Stockham chain of csrc/fft_lds.hpp (any_plan / any_fft), replayed in NumPy: same factorisation, same stage formula,
same split into butterfly stages (radices 2, 3, 4, 5, 7) and one-output-at-a-time stages (larger primes), same index arithmetic into ONE twiddle table — checked against
numpy.fft for the lengths the any-size kernels meet (prime factors 2..157, odd and prime lengths).  Runs without a GPU; the device code is
checked against the oracle in tests/test_kdyn_gpu.py, test_sh23_gpu.py, test_shb23_gpu.py."""
import numpy as np
import pytest


def any_plan(L):
    """csrc/fft_lds.hpp any_plan: radices 4, then 2, then the odd primes in ascending order."""
    r, n = [], L
    while n % 4 == 0:
        r.append(4); n //= 4
    if n % 2 == 0:
        r.append(2); n //= 2
    f = 3
    while n > 1:
        while n % f == 0:
            r.append(f); n //= f
        f += 2
    return r


def any_fft(x, inverse):
    """Stage invariant n * s == L:  y[q + s (R p + j)] = w_n^{p j} sum_k x[q + s (p + k n/R)] w_R^{j k}, every w taken from tw[k] = exp(-2 pi i k / L).
    Radices 2, 3, 4, 5, 7: one butterfly at a time (R inputs -> R outputs); larger primes: one output at a time as a direct sum."""
    L = len(x)
    tw = np.exp(-2j * np.pi * np.arange(L) / L)
    w = (lambda i: np.conj(tw[i])) if inverse else (lambda i: tw[i])
    src = np.array(x, dtype=complex)
    n, s = L, 1
    for R in any_plan(L):
        M, wstep, xs = n // R, L // R, s * (n // R)
        dst = np.empty(L, dtype=complex)
        if R <= 5 or R == 7:
            for jj in range(L // R):
                p, q = jj // s, jj % s
                v = np.array([src[q + s * p + k * xs] for k in range(R)])
                out = np.array([sum(v[k] * w(((j * k) % R) * wstep) for k in range(R)) for j in range(R)])      # Butterfly<R>
                for k in range(R):
                    if k and M > 1:
                        assert p * s * k < L
                        out[k] = out[k] * w(p * s * k)
                    dst[q + s * R * p + s * k] = out[k]
        else:
            for o in range(L):
                q, rj = o % s, o // s
                j, p = rj % R, rj // R
                base = q + s * p
                acc, e = src[base], 0
                for k in range(1, R):
                    e += j
                    if e >= R:
                        e -= R
                    acc = acc + src[base + k * xs] * w(e * wstep)
                if M > 1 and j:
                    assert p * s * j < L
                    acc = acc * w(p * s * j)
                dst[o] = acc
        src, n, s = dst, M, s * R
    assert n == 1 and s == L
    return src


@pytest.mark.parametrize("L", [2, 3, 4, 5, 8, 9, 15, 21, 33, 37, 64, 66, 97, 111, 127, 132, 186, 195, 222, 250, 333, 465, 471])
def test_stockham_chain_of_any_length(L):
    assert int(np.prod(any_plan(L))) == L
    rs = np.random.RandomState(L)
    x = rs.standard_normal(L) + 1j * rs.standard_normal(L)
    tol = 1e-13 * max(1.0, np.sqrt(L)) * np.abs(x).sum()
    assert np.abs(any_fft(x, False) - np.fft.fft(x)).max() < tol
    assert np.abs(any_fft(x, True) - np.fft.ifft(x) * L).max() < tol


@pytest.mark.parametrize("N", [5, 8, 12, 33, 50])
def test_makhoul_full_length_dct_pair(N):
    """The SHB23 any-N form (csrc/shb23.hip, dct2<0> / dct3<0>): DCT-II and DCT-III of any length through ONE complex transform of the full
    length, against the defining sums."""
    rs = np.random.RandomState(N)
    x = rs.standard_normal(N)
    n, k = np.arange(N), np.arange(N)
    tw4 = np.exp(-1j * np.pi * k / (2 * N))
    perm = np.where(n & 1, N - 1 - (n >> 1), n >> 1)             # v[perm[n]] = x[n]
    v = np.zeros(N, dtype=complex); v[perm] = x
    y2 = 2.0 * np.real(any_fft(v, False) * tw4)
    ref2 = 2.0 * (np.cos(np.pi * np.outer(k, 2 * n + 1) / (2 * N)) @ x)
    assert np.abs(y2 - ref2).max() < 1e-12 * np.abs(x).sum()
    a = np.where(k == 0, 1.0, 2.0) * x
    y3 = np.real(any_fft(a * np.conj(tw4), True))[perm]
    ref3 = x[0] + 2.0 * (np.cos(np.pi * np.outer(2 * n + 1, k[1:]) / (2 * N)) @ x[1:])
    assert np.abs(y3 - ref3).max() < 1e-12 * np.abs(x).sum()
