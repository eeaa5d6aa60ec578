"""Random-walk corpus and negative-sampling loss on the GPU vs the reference's outputs / rules."""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import load_golden, csr_from

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _upload(csr):
    csr = sp.csr_matrix(csr); csr.sort_indices()
    return (torch.from_numpy(csr.indptr.astype(np.int32)).to(DEV), torch.from_numpy(csr.indices.astype(np.int32)).to(DEV),
            torch.from_numpy(csr.data.astype(np.float32)).to(DEV))


def test_deterministic_graph_matches_reference_files_exactly():
    """on a perfect matching the reference's walks are deterministic: pair matrix and negative table must be identical."""
    from ctgcn_amd.walks import random_walk_corpus, negative_table
    g = load_golden("negloss.npz")
    n = len(g["match_adj_indptr"]) - 1
    L, W = [int(x) for x in g["match_LW"]]
    for per_round in (10, 3):
        pairs, freq = random_walk_corpus(*_upload(csr_from(g, "match_adj", n)), walk_length=L, walk_time=W, weighted=True, seed=5,
                                         walks_per_round=per_round)
        got = pairs.to_scipy(); got.sort_indices()
        want = csr_from(g, "match_pairs", n)
        assert np.array_equal(got.indptr, want.indptr) and np.array_equal(got.indices, want.indices)
        assert np.array_equal(negative_table(freq), g["match_neg"])


def test_corpus_structure_and_statistics_on_uci_snapshot():
    """every pair is a pair the reference could emit (within walk_length hops), the matrix is symmetric with zero diagonal,
    frequencies count exactly the emitted events, and the negative table follows the reference run's distribution."""
    from ctgcn_amd.utils import symmetric_csr_from_rows
    from ctgcn_amd.walks import random_walk_corpus, negative_table
    snaps = load_golden("uci_snapshots.npz")
    g = load_golden("negloss.npz")
    n = len(snaps["node_names"])
    adj = symmetric_csr_from_rows(snaps["t6_src"], snaps["t6_dst"], snaps["t6_w"], n)
    pairs, freq = random_walk_corpus(*_upload(adj), walk_length=5, walk_time=10, weighted=True, seed=1)
    m = pairs.to_scipy()
    assert (m != m.T).nnz == 0 and m.diagonal().sum() == 0
    reach = sp.identity(n, format="csr")
    hop = (adj != 0).astype(np.float64) + sp.identity(n, format="csr")
    for _ in range(5):
        reach = ((reach @ hop) != 0).astype(np.float64)
    assert (m.multiply(reach) != m).nnz == 0                                  # no pair beyond 5 hops
    deg = np.diff(adj.indptr)
    assert np.all(np.diff(m.indptr)[deg == 0] == 0)                            # isolated nodes have no partners
    f = freq.cpu().numpy()
    assert f.sum() % 2 == 0 and np.all(f[deg == 0] == 0)
    # distribution vs the reference's own seeded run (same graph, same walk_time): table sizes and per-node counts agree
    counts = np.bincount(negative_table(freq), minlength=n)
    ref_counts = g["uci_neg_counts"].astype(np.float64)
    assert abs(counts.sum() - ref_counts.sum()) <= 0.05 * ref_counts.sum()
    assert np.corrcoef(counts, ref_counts)[0, 1] > 0.98
    assert abs(m.nnz - int(g["uci_pairs_nnz"])) <= 0.1 * int(g["uci_pairs_nnz"])
    assert np.corrcoef(np.diff(m.indptr), g["uci_pairs_rowcount"])[0, 1] > 0.95
    # same seed -> same corpus; other seed -> different walks
    p2, f2 = random_walk_corpus(*_upload(adj), walk_length=5, walk_time=10, weighted=True, seed=1)
    assert torch.equal(p2.col, pairs.col) and torch.equal(f2, freq)
    p3, _ = random_walk_corpus(*_upload(adj), walk_length=5, walk_time=10, weighted=True, seed=2)
    assert p3.col.numel() != pairs.col.numel() or not torch.equal(p3.col, pairs.col)


def test_weighted_walks_follow_the_edge_weights():
    """star with one heavy spoke: first hops from the hub must split ~ proportionally to the weights."""
    from ctgcn_amd.walks import random_walk_corpus
    n = 6
    w = np.array([8.0, 1.0, 1.0, 1.0, 1.0])
    adj = sp.coo_matrix((np.concatenate([w, w]), (np.concatenate([np.zeros(5, int), np.arange(1, 6)]),
                                                  np.concatenate([np.arange(1, 6), np.zeros(5, int)]))), shape=(n, n)).tocsr()
    _, freq = random_walk_corpus(*_upload(adj), walk_length=1, walk_time=6000, weighted=True, seed=3)
    f = freq.cpu().numpy().astype(np.float64)
    # walks of one step: hub walks pick spoke i with prob w_i/12; spoke walks always go to the hub (1 event each)
    hub_picks = f[1:] - 6000.0
    np.testing.assert_allclose(hub_picks / hub_picks.sum(), w / w.sum(), atol=0.02)
    _, freq_u = random_walk_corpus(*_upload(adj), walk_length=1, walk_time=6000, weighted=False, seed=3)
    hp = freq_u.cpu().numpy()[1:] - 6000.0
    np.testing.assert_allclose(hp / hp.sum(), np.full(5, 0.2), atol=0.02)


def test_negative_sampling_draws_follow_reference_rules():
    from ctgcn_amd.metrics import NegativeSamplingLoss
    rng = np.random.default_rng(0)
    n, num = 200, 6
    rows = np.empty(n, dtype=object)
    for i in range(n):
        rows[i] = sorted(rng.choice(n, size=int(rng.integers(0, 15)), replace=False).tolist())
    table = rng.integers(0, n, 500).tolist()
    loss = NegativeSamplingLoss([rows], [table], neg_num=num, Q=3)
    batch = torch.from_numpy(rng.choice(n, 64, replace=False)).to(DEV)
    seen = {}
    for rep in range(200):
        cnt, ni, pi, gi = loss.sample_indices(0, batch)
        ni, pi, gi = ni.cpu().numpy(), pi.cpu().numpy(), gi.cpu().numpy()
        assert cnt == len(ni) == sum(min(len(rows[b]), num) for b in batch.cpu().numpy())
        assert len(gi) == num and set(gi) <= set(table)
        for b in batch.cpu().numpy():
            got = pi[ni == b]
            assert len(got) == min(len(rows[b]), num) and len(set(got)) == len(got) and set(got) <= set(rows[b])
            if len(rows[b]) <= num:
                assert sorted(got) == rows[b]
            else:
                for x in got:
                    seen[(b, x)] = seen.get((b, x), 0) + 1
    # uniform without replacement: each partner of a long list is picked with probability num/deg
    for b in batch.cpu().numpy():
        if len(rows[b]) > num:
            p = num / len(rows[b])
            for x in rows[b]:
                assert abs(seen.get((b, x), 0) / 200.0 - p) < 0.17
    empty = NegativeSamplingLoss([np.array([[] for _ in range(n)], dtype=object)], [table], neg_num=num)
    assert empty.sample_indices(0, batch)[0] == 0


def test_negative_sampling_loss_matches_reference_value_and_gradients():
    from ctgcn_amd.metrics import NegativeSamplingLoss
    g = load_golden("negloss.npz")
    n2, T, dim, neg_num, Q = [int(x) for x in g["loss_cfg"]]
    pair_lists, tables, embs = [], [], []
    for t in range(T):
        lens, flat = g["loss_pairs%d_len" % t], g["loss_pairs%d_flat" % t]
        ptr = np.concatenate([[0], np.cumsum(lens)])
        rows = np.empty(n2, dtype=object)
        for i in range(n2):
            rows[i] = flat[ptr[i]:ptr[i + 1]].tolist()
        pair_lists.append(rows)
        tables.append(g["loss_table%d" % t].tolist())
        embs.append(torch.from_numpy(g["loss_emb%d" % t]).to(DEV).requires_grad_(True))
    loss = NegativeSamplingLoss(pair_lists, tables, neg_num=neg_num, Q=Q)([embs, torch.from_numpy(g["loss_batch"]).to(DEV)])
    assert loss.shape == (1,)
    np.testing.assert_allclose(loss.detach().cpu().numpy(), g["loss_value"], rtol=1e-5)
    loss.backward()
    for t in range(T):
        np.testing.assert_allclose(embs[t].grad.cpu().numpy(), g["loss_grad%d" % t], rtol=1e-4, atol=1e-6)


def test_walk_generator_writes_reference_file_layout(tmp_path):
    import ctgcn_amd
    from ctgcn_amd.preprocessing import WalkGenerator
    snaps = load_golden("uci_snapshots.npz")
    names = [str(x) for x in snaps["node_names"]]
    os.makedirs(tmp_path / "1.format"); os.makedirs(tmp_path / "nodes_set")
    (tmp_path / "nodes_set" / "nodes.csv").write_text("\n".join(names) + "\n")
    for t in (5, 6):
        with open(tmp_path / "1.format" / str(snaps["files"][t]), "w") as fp:
            fp.write("from_id\tto_id\tweight\n")
            for s, d, w in zip(snaps["t%d_src" % t], snaps["t%d_dst" % t], snaps["t%d_w" % t]):
                fp.write("%s\t%s\t%d\n" % (names[s], names[d], int(w)))
    gen = WalkGenerator(str(tmp_path), "1.format", "walk_pairs", "node_freq", "nodes_set/nodes.csv", walk_time=4, walk_length=3)
    gen.get_walk_info_all_time()
    assert sorted(os.listdir(tmp_path / "walk_pairs")) == ["2004-09.npz", "2004-10.npz"]
    assert sorted(os.listdir(tmp_path / "node_freq")) == ["2004-09.json", "2004-10.json"]
    m = sp.load_npz(str(tmp_path / "walk_pairs" / "2004-10.npz"))
    assert sp.isspmatrix_coo(m) and m.shape == (1899, 1899) and set(np.unique(m.data)) == {1.0}
    table = json.load(open(tmp_path / "node_freq" / "2004-10.json"))
    assert isinstance(table, list) and all(isinstance(x, int) for x in table[:10]) and table == sorted(table)
    dl = ctgcn_amd.DataLoader(names, 2, has_cuda=True)
    pairs = dl.get_node_pair_list(str(tmp_path / "walk_pairs"), 0, 2)
    freqs = dl.get_node_freq_list(str(tmp_path / "node_freq"), 0, 2)
    assert len(pairs) == 2 and len(pairs[1]) == 1899 and freqs[1].tolist() == table
    assert pairs[1][0] == sorted(m.tocsr()[0].indices.tolist())
