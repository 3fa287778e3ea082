"""CHECKER (test infrastructure, see oracle/__init__.py) of ``preprocess_aggregate`` /
``aggregate_neighbors`` (call sites /root/reference/model_pn.py:257-264,
test_neighbor_aggregation.py:75-98): the dense torch statement of the definition that
pigs_amd/csrc/aggregate.hip implements sparsely.  Only tests and tools import it.

PARITY UNPINNED.  The arithmetic of these two methods exists only in the reference's absent
CUDA source; the call sites fix the signature, the shapes (features [N,L], transform [L,L],
queries/keys [N,K], frequencies [F], distance_transform [L,2E] with E = 2 d F + 1 -> [N,L]),
the dtype (float64 in the reference's gradcheck) and that the result is differentiable wrt all
six arguments -- nothing else.  The definition below is this repo's own, chosen to be the
natural neighbour attention over the structure ``preprocess`` already has:

* neighbours of Gaussian i = the Gaussians j whose q <= q_max ellipse reaches the centre of i
  (the sampler's own cut-off, evaluated at samples = means as model_pn.py:648 sets it up);
* weight a_ij = softmax over the neighbours j of <queries_i, keys_j> / sqrt(K);
* message m_ij = transform @ features_j + distance_transform @ [e_ij ; g_ij e_ij], with
  e_ij = (sin(f_k dx), cos(f_k dx), sin(f_k dy), cos(f_k dy) for k < F, 1) the Fourier embedding
  of mu_j - mu_i and g_ij = exp(-q_ij / 2) the density of Gaussian j at the centre of i;
* out_i = sum_j a_ij m_ij.

Dense on purpose (a checker: [N,N] mask, [N,N,2E] embedding): differentiable torch tensor ops in
float32 or float64 on any device.
"""
import math

import torch


def neighbor_structure(means, conics_flat, q_max):
    """Dense neighbour relation.  Returns (mask [N,N] bool: j is a neighbour of i,
    delta [N,N,d] = mu_j - mu_i, g [N,N] = exp(-q_ij/2)); constants for autograd."""
    with torch.no_grad():
        N, d = means.shape
        delta = means[None, :, :] - means[:, None, :]                    # [i, j, d] = mu_j - mu_i
        if d == 1:
            q = conics_flat.reshape(1, N) * delta[..., 0] ** 2
        else:
            a, b, c = conics_flat[:, 0], conics_flat[:, 1], conics_flat[:, 2]
            dx, dy = delta[..., 0], delta[..., 1]
            q = a[None] * dx * dx + 2 * b[None] * dx * dy + c[None] * dy * dy   # conic of j
        return q <= q_max, delta, torch.exp(-0.5 * q)


def aggregate(mask, delta, g, features, transform, queries, keys, frequencies, distance_transform):
    N, d = delta.shape[0], delta.shape[2]
    L, K, F = features.shape[1], queries.shape[1], frequencies.shape[0]
    E = 2 * d * F + 1
    if transform.shape != (L, L) or keys.shape != (N, K) or distance_transform.shape != (L, 2 * E):
        raise ValueError(f"aggregate_neighbors: expected transform [{L},{L}], keys [{N},{K}], "
                         f"distance_transform [{L},{2 * E}] (E = 2*d*F + 1 = {E})")
    scores = (queries @ keys.t()) / math.sqrt(K)                          # [i, j]
    scores = scores.masked_fill(~mask, float("-inf"))
    attn = torch.softmax(scores, dim=1)                                   # rows always contain j = i
    phase = delta[..., None] * frequencies                                # [i, j, d, F]
    emb = torch.stack((torch.sin(phase), torch.cos(phase)), dim=-1)       # [i, j, d, F, 2]
    emb = emb.permute(0, 1, 3, 2, 4).reshape(N, N, 2 * d * F)             # (k, axis, sin|cos) order
    emb = torch.cat((emb, torch.ones((N, N, 1), dtype=emb.dtype, device=emb.device)), dim=-1)  # [i, j, E]
    emb2 = torch.cat((emb, g[..., None] * emb), dim=-1)                   # [i, j, 2E]
    msg_feat = features @ transform.t()                                   # [j, L]
    out = attn @ msg_feat + torch.einsum("ij,ije,le->il", attn, emb2, distance_transform)
    return out
