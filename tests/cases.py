"""Input builders shared by oracle/make_goldens.py (reference side) and the parity tests.

Large inputs are regenerated from fsw_gnn_amd/synth.py (counter-based, identical on every machine);
the golden fixtures hold only expected outputs, sampled rows and norms for them.
"""
import numpy as np

from fsw_gnn_amd import synth


def random_freqs(n, seed):
    """The reference's freqs_init='random' law u/(1-u) (fsw_embedding.py:519-526) on synth uniforms."""
    u = synth.uniform01(seed, 601, 0, n)
    u.sort()
    return (u / (1 - u)).astype(np.float32)


def conv_params(in_ch, out_ch, embed_dim, seed):
    """projVecs / 'spread' freqs / one Linear layer (fan_in = embed_dim + in_ch) for an FSW_conv."""
    S = embed_dim - 1
    V = synth.unit_slices(S, in_ch, seed=seed)
    fr = synth.spread_freqs(S)
    fan_in = embed_dim + in_ch
    Wl = (synth.normal(seed + 1, 1, (out_ch, fan_in), dtype=np.float64) / np.sqrt(fan_in)).astype(np.float32)
    bl = (0.1 * synth.normal(seed + 2, 1, (out_ch,), dtype=np.float64)).astype(np.float32)
    return V, fr, Wl, bl


def pointcloud_1k():
    """BASELINE config 1: 1k points x 64 dims x 32 slices, unit weights."""
    return dict(X=synth.features(1000, 64, seed=11), V=synth.unit_slices(32, 64, seed=12),
                freqs=synth.spread_freqs(32))


def conv10k():
    """BASELINE config 2: 10k nodes / 100k edges, 64 feat, 128 slices (+ degree column)."""
    n, E, d, out_ch, embed_dim = 10_000, 100_000, 64, 64, 129
    V, fr, Wl, bl = conv_params(d, out_ch, embed_dim, seed=43)
    return dict(n=n, d=d, out_ch=out_ch, embed_dim=embed_dim, edge_index=synth.er_multigraph(n, E, seed=41),
                X=synth.features(n, d, seed=42), V=V, freqs=fr, lin_w=Wl, lin_b=bl)


def rmat(scale=14, edge_factor=16, d=32, S=64):
    """Reduced-scale version of BASELINE config 5 (RMAT, skewed degrees)."""
    n = 1 << scale
    return dict(n=n, d=d, S=S, edge_index=synth.rmat_graph(scale, edge_factor * n, seed=51),
                X=synth.features(n, d, seed=52), V=synth.unit_slices(S, d, seed=53), freqs=synth.spread_freqs(S))


def er1m(nslices=256):
    """BASELINE config 3: 1M nodes / 10M edges, 128 feat, 256 slices (+ degree column)."""
    n, E, d = 1_000_000, 10_000_000, 128
    V, fr, Wl, bl = conv_params(d, 128, nslices + 1, seed=73)
    return dict(n=n, d=d, out_ch=128, embed_dim=nslices + 1, edge_index=synth.er_multigraph(n, E, seed=71),
                X=synth.features(n, d, seed=72), V=V, freqs=fr, lin_w=Wl, lin_b=bl)
