"""Multi-GPU: shard the slice axis, reassemble with ONE all-gather (RCCL over xGMI; gloo in CPU tests).

The reference has no distributed code at all; this is the partition SURVEY.md 8(e) prescribes.
Every (recipient, slice) pair is independent and slice k is paired with frequency k, so rank r of G owns a
contiguous block of slices -- rows [ka, kb) of projVecs and entries [ka, kb) of freqs.  X, the CSR adjacency
and the total-mass column are replicated (each rank builds the CSR itself).  Each rank runs the identical
kernels on its block and contributes local[n, 1 + (kb - ka)]; one all_gather_into_tensor collects
[G, n, 1 + width] and a local strided copy interleaves the blocks into out[n, d_out].  There is no reduction,
so the result is bit-identical to the single-GPU result.
"""
import torch
import torch.distributed as dist


def slice_partition(num_slices, world_size):
    """Contiguous, balanced blocks: returns [(ka, kb)] * world_size (first `rem` ranks get one extra slice)."""
    base, rem = divmod(num_slices, world_size)
    out, a = [], 0
    for r in range(world_size):
        b = a + base + (1 if r < rem else 0)
        out.append((a, b))
        a = b
    return out


def node_block(num_rows, world_size, rank):
    """Recipient-row sharding (FSW_conv.enable_node_parallel): rank r owns rows [r0, r0 + nl) of equal-size blocks of
    per = ceil(num_rows / world_size) rows (the last blocks may be short or empty).  Returns (per, r0, nl)."""
    per = -(-num_rows // world_size)
    r0 = min(rank * per, num_rows)
    return per, r0, min(r0 + per, num_rows) - r0


def all_gather_slice_blocks(local, parts, has_mass, out, group=None):
    """local [n, has_mass + max_width] of this rank -> out[:, :has_mass + S] on every rank.

    `parts` is slice_partition(S, world).  Blocks are padded to the widest block so that one
    all_gather_into_tensor (a single collective, equal message sizes) suffices.
    """
    world = dist.get_world_size(group)
    n = local.shape[0]
    wmax = max(b - a for a, b in parts)
    assert local.shape[1] == has_mass + wmax and local.is_contiguous()
    flat = torch.empty((world * n, has_mass + wmax), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(flat, local, group=group)      # concatenation along dim 0 (valid for RCCL and gloo)
    gathered = flat.view(world, n, has_mass + wmax)
    if has_mass:
        out[:, 0] = gathered[0, :, 0]            # every rank computed the same total-mass column
    for r, (a, b) in enumerate(parts):
        if b > a:
            out[:, has_mass + a:has_mass + b] = gathered[r, :, has_mass:has_mass + (b - a)]
    return out


def sharded_embed_into(emb_mod, X, graph, out, out_scale=1.0, group=None, x_copy=None):
    """Slice-sharded version of FSW_embedding.embed_into: every rank ends with the full embedding in `out`."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return emb_mod.embed_into(X, graph, out, out_scale=out_scale, x_copy=x_copy)
    has_mass = 1 if emb_mod.encode_total_mass else 0
    parts = slice_partition(emb_mod.nSlices, world)
    wmax = max(b - a for a, b in parts)
    ka, kb = parts[rank]
    # columns past this rank's block (only when S is not divisible by the world size) are never read back
    local = torch.empty((graph.num_rows, has_mass + wmax), dtype=X.dtype, device=X.device)
    if kb > ka:
        emb_mod.embed_into(X, graph, local, out_scale=out_scale, slice_range=(ka, kb), x_copy=x_copy)
    elif x_copy is not None:
        x_copy.copy_(X)
    return all_gather_slice_blocks(local, parts, has_mass, out, group)
