"""Mutual-coherence minimisation of the slice directions (init-time, host logic in torch).

Restates the reference's minimize_mutual_coherence (reference fsw_embedding.py:3045-3248): projected gradient
descent on the unit sphere rows of an l_p surrogate of the largest off-diagonal Gram entry, with p continued
from 3 to 1e13 and the step size carried from one stage to the next.  It runs once when an embedding is built with
minimize_slice_coherence=True (always, for FSW_conv: reference fsw_conv.py:321) and is not on the forward path.

    G(X)   = X X^T with zeroed diagonal,   mu(X) = max |G|
    obj_p  = mu * ( rho * sum |G / mu|^p )^(1/p),   rho = 1 / (2 n (n-1))
    step   : X <- normalise_rows( X - t * grad obj_p ), accepted only when obj_p decreases
    t      : grown by 2x while the very first steps keep improving (best one kept), then halved on every rejection;
             a stage ends after 5 consecutive improvements below 1e-4 (relative to 1 - obj), at t < 1e-5 or after
             1000 iterations, and is kept only if mu itself went down.
"""
import torch

P_SCHEDULE = (3, 6, 10, 20, 50, 100, 200, 500, 1000, 2000, 5000, 1e4, 2e4, 5e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13)
STEP_INIT, STEP_MIN, STEP_MAX, STEP_SHRINK = 2000.0, 1e-5, 1e10, 0.5
MAX_ITERS, LOW_IMPROVEMENT, MAX_LOW_STRIKES = 1000, 1e-4, 5


def gram_offdiag(X):
    G = X @ X.t()
    G.fill_diagonal_(0)
    return G


def mutual_coherence(X):
    """Largest absolute inner product between two different (unit) rows."""
    return gram_offdiag(torch.nn.functional.normalize(X, p=2, dim=1, eps=0)).abs().max()


class _State:
    """A point on the product of spheres with its Gram matrix, coherence and surrogate objective."""

    def __init__(self, X, p):
        n = X.shape[0]
        self.X = X
        self.G = gram_offdiag(X)
        self.mu = self.G.abs().max()
        self.obj = self.mu * torch.pow(torch.sum(torch.pow((self.G / self.mu).abs(), p)) / (2.0 * n * (n - 1.0)), 1.0 / p)

    def gradient(self, p):
        n = self.X.shape[0]
        Gn = self.G / self.mu                                   # largest entry has magnitude 1: |Gn|^p cannot overflow
        A = Gn.abs()
        total = torch.sum(torch.pow(A, p))
        lead = (2.0 * n * (n - 1.0)) ** (-1.0 / p) / torch.pow(total, 1.0 - 1.0 / p)
        pull = (torch.pow(A, p - 1.0) * torch.sign(Gn)) @ self.X
        radial = (torch.pow(A, p) @ (self.mu * torch.ones((n, 1), dtype=self.X.dtype, device=self.X.device))) * self.X
        return lead * (pull - radial)

    def moved(self, step, p):
        return _State(torch.nn.functional.normalize(self.X - step * self.gradient(p), p=2, dim=1, eps=0), p)


def _one_stage(X0, p, step, report):
    p = float(p)
    n = X0.shape[0]
    if X0.numel() == 0:
        return X0, step
    if n == 1:
        return torch.nn.functional.normalize(X0), step
    cur = _State(X0, p)
    mu_start, step_start = cur.mu, step
    seeking = True                    # still growing the very first step size
    best_seek, best_seek_obj, best_seek_step = None, float('inf'), step
    strikes = 0
    for it in range(1, MAX_ITERS + 1):
        cand = cur.moved(step, p)
        if not bool(cand.obj < cur.obj):                        # rejected
            if seeking:
                seeking, step = False, best_seek_step           # stop growing: fall back to the best step tried
            else:
                if step * STEP_SHRINK < STEP_MIN:
                    break
                step *= STEP_SHRINK
            continue
        if seeking:
            if bool(cand.obj < best_seek_obj) and step / STEP_SHRINK <= STEP_MAX:
                best_seek, best_seek_obj, best_seek_step = cand, float(cand.obj), step
                step = step / STEP_SHRINK                       # try a larger step from the SAME point
                continue
            seeking, step, cand = False, best_seek_step, best_seek
        gain = float((cur.obj - cand.obj) / (1.0 - cur.obj))    # relative to 1 - obj: informative near 1
        cur = cand
        if report:
            print('#%.2d  surrogate incoherence %g  step %g  improvement %g' % (it, 1.0 - float(cur.obj), step, gain))
        if gain <= LOW_IMPROVEMENT:
            strikes += 1
            if strikes >= MAX_LOW_STRIKES:
                break
        else:
            strikes = 0
    if bool(cur.mu < mu_start):
        return cur.X, step
    return X0, step_start                                        # mu did not improve: revert point and step size


def minimize_mutual_coherence(X_init, report=False):
    """Rows of X_init [n, d] -> unit rows with (locally) minimal mutual coherence; same schedule as the reference."""
    X = torch.nn.functional.normalize(X_init, p=2, dim=1, eps=0)
    step = STEP_INIT
    for p in P_SCHEDULE:
        X, step = _one_stage(X, p, step, report)
        if report:
            print('p = %g: incoherence %g' % (p, 1.0 - float(gram_offdiag(X).abs().max())))
    return X
