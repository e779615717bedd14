"""Counter-based synthetic inputs (graphs, features, slices, frequencies).

Everything here is a pure function of (seed, stream, index) through a splitmix64 hash, evaluated with
numpy uint64 arithmetic.  The same call therefore yields the same graph / features in the build
container (where the goldens are captured from the reference) and on the GPU box (where the HIP path
is checked against them), independent of torch's RNG state or version.

The shapes follow SURVEY.md section 8(d):
  * point cloud  X ~ N(0,1) [n, d]
  * ER-style directed multigraph: src, dst ~ U{0..n-1} i.i.d.  (duplicates and isolated nodes occur)
  * RMAT (Graph500 a,b,c = .57,.19,.19)
  * projVecs = normalised N(0,1) rows, freqs = the reference's 'spread' formula
    (/root/reference/fsw_embedding.py:529-531).
"""
import numpy as np

_U64 = np.uint64
_MASK53 = _U64((1 << 53) - 1)
_GOLDEN = _U64(0x9E3779B97F4A7C15)
_CHUNK = 1 << 24


def splitmix64(x):
    """One splitmix64 finalisation round on a uint64 array (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = (x + _GOLDEN).astype(_U64)
        z = (z ^ (z >> _U64(30))) * _U64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> _U64(27))) * _U64(0x94D049BB133111EB)
        z = z ^ (z >> _U64(31))
    return z


def _hash(seed, stream, idx):
    """hash(seed, stream, idx) -> uint64; idx is a uint64 array of counters."""
    with np.errstate(over="ignore"):
        key = splitmix64(np.asarray([seed], dtype=_U64) * _U64(0xD1342543DE82EF95) + _U64(stream))
        return splitmix64(splitmix64(idx ^ key[0]) + key[0])


def uniform01(seed, stream, start, count):
    """count doubles in the open interval (0,1), counters start..start+count-1."""
    idx = np.arange(start, start + count, dtype=_U64)
    h = _hash(seed, stream, idx)
    return ((h >> _U64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def randint(seed, stream, high, count, start=0):
    """count int64 draws from U{0..high-1} (modulo reduction; bias < 2^-40 for high < 2^24)."""
    out = np.empty(count, dtype=np.int64)
    for a in range(0, count, _CHUNK):
        b = min(count, a + _CHUNK)
        idx = np.arange(start + a, start + b, dtype=_U64)
        out[a:b] = (_hash(seed, stream, idx) % _U64(high)).astype(np.int64)
    return out


def normal(seed, stream, shape, dtype=np.float32):
    """N(0,1) array of the given shape (Box-Muller on counter pairs), generated in float64."""
    count = int(np.prod(shape))
    out = np.empty(count, dtype=dtype)
    for a in range(0, count, _CHUNK):
        b = min(count, a + _CHUNK)
        idx = np.arange(a, b, dtype=_U64)
        h1 = _hash(seed, stream, idx * _U64(2))
        h2 = _hash(seed, stream, idx * _U64(2) + _U64(1))
        u1 = ((h1 >> _U64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)
        u2 = ((h2 >> _U64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)
        out[a:b] = (np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)).astype(dtype)
    return out.reshape(shape)


def er_multigraph(n, num_edges, seed):
    """edge_index [2, E] int64 with row 0 = source (sender), row 1 = target (recipient)."""
    ei = np.empty((2, num_edges), dtype=np.int64)
    ei[0] = randint(seed, 101, n, num_edges)
    ei[1] = randint(seed, 102, n, num_edges)
    return ei


def rmat_graph(scale, num_edges, seed, a=0.57, b=0.19, c=0.19):
    """Graph500-style RMAT edge list on 2^scale vertices, edge_index [2, E] int64."""
    src = np.zeros(num_edges, dtype=np.int64)
    dst = np.zeros(num_edges, dtype=np.int64)
    ab, abc = a + b, a + b + c
    for lo in range(0, num_edges, _CHUNK):
        hi = min(num_edges, lo + _CHUNK)
        s = np.zeros(hi - lo, dtype=np.int64)
        t = np.zeros(hi - lo, dtype=np.int64)
        for bit in range(scale):
            u = uniform01(seed, 200 + bit, lo, hi - lo)
            sbit = (u >= ab).astype(np.int64)                       # quadrants c, d -> source bit 1
            tbit = (((u >= a) & (u < ab)) | (u >= abc)).astype(np.int64)  # quadrants b, d -> target bit 1
            s |= sbit << bit
            t |= tbit << bit
        src[lo:hi] = s
        dst[lo:hi] = t
    return np.stack([src, dst], axis=0)


def unit_slices(num_slices, d, seed, dtype=np.float32):
    """projVecs [S, d]: N(0,1) rows normalised in float64 (reference: fsw_embedding.py:455-456)."""
    v = normal(seed, 301, (num_slices, d), dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    return v.astype(dtype)


def spread_freqs(num_freqs, dtype=np.float32):
    """The reference's freqs_init='spread' (fsw_embedding.py:529-531), computed in float64."""
    f = (0.5 + np.arange(num_freqs, dtype=np.float64)) / num_freqs
    return (f / (1.0 - f)).astype(dtype)


def features(n, d, seed, dtype=np.float32):
    return normal(seed, 401, (n, d), dtype=dtype)


def edge_weights(num_edges, seed, dtype=np.float32):
    """Positive non-unit edge weights in (0.05, 1.05)."""
    return (uniform01(seed, 501, 0, num_edges) + 0.05).astype(dtype)
