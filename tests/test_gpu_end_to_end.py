"""The reference's whole CTGCN-C workflow on the bundled UCI snapshots, every stage through this package:
preprocessing (k-core files + walk corpus) -> loaders -> CTGCN-C -> NegativeSamplingLoss -> Adam -> TSV export."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_ctgcn_c_workflow_on_uci(tmp_path):
    import ctgcn_amd
    from ctgcn_amd import export
    from ctgcn_amd.metrics import NegativeSamplingLoss
    from ctgcn_amd.preprocessing import StructureInfoGenerator, WalkGenerator
    snaps = load_golden("uci_snapshots.npz")
    names = [str(x) for x in snaps["node_names"]]
    os.makedirs(tmp_path / "1.format"); os.makedirs(tmp_path / "nodes_set")
    (tmp_path / "nodes_set" / "nodes.csv").write_text("\n".join(names) + "\n")
    files = [str(f) for f in snaps["files"]]
    for t, f in enumerate(files):
        with open(tmp_path / "1.format" / f, "w") as fp:
            fp.write("from_id\tto_id\tweight\n")
            for s, d, w in zip(snaps["t%d_src" % t], snaps["t%d_dst" % t], snaps["t%d_w" % t]):
                fp.write("%s\t%s\t%d\n" % (names[s], names[d], int(w)))
    # preprocessing task (reference main.py --task preprocessing)
    StructureInfoGenerator(str(tmp_path), "1.format", "2.core", "nodes_set/nodes.csv").get_kcore_graph_all_time()
    WalkGenerator(str(tmp_path), "1.format", "walk_pairs", "node_freq", "nodes_set/nodes.csv", walk_time=10, walk_length=5).get_walk_info_all_time()
    # embedding task (reference train.gnn_embedding, one window of 3 snapshots)
    start, dur, n = 4, 3, len(names)
    dl = ctgcn_amd.DataLoader(names, len(files), has_cuda=True)
    adj_list = dl.get_core_adj_list(str(tmp_path / "2.core"), start, dur, max_core=-1)
    x_list, in_dim = dl.get_feature_list(None, start, dur)
    pairs = dl.get_node_pair_list(str(tmp_path / "walk_pairs"), start, dur)
    freqs = dl.get_node_freq_list(str(tmp_path / "node_freq"), start, dur)
    assert in_dim == n and [len(a) for a in adj_list] == [4, 3, 2]
    torch.manual_seed(0)
    model = ctgcn_amd.CTGCN(in_dim, 128, 128, 1, 2, dur, rnn_type="GRU", model_type="C", trans_activate_type="L").to(DEV)
    loss_model = NegativeSamplingLoss(pairs, freqs, neg_num=20, Q=20, seed=7)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    batch = torch.arange(n, device=DEV)
    losses = []
    for epoch in range(6):
        opt.zero_grad()
        out = model(x_list, adj_list)                     # [T, N, 128]
        loss = loss_model([list(out), batch])
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    with torch.no_grad():
        emb = model(x_list, adj_list)
    export.save_embedding(emb, files, start, str(tmp_path / "emb"), names)
    assert sorted(os.listdir(tmp_path / "emb")) == ["2004-08.csv", "2004-09.csv", "2004-10.csv"]
    import pandas as pd
    df = pd.read_csv(tmp_path / "emb" / "2004-10.csv", sep="\t", index_col=0)       # how evaluation/*.py reads it
    assert df.shape == (n, 128) and list(df.index[:3]) == names[:3] and list(df.columns[:2]) == ["0", "1"]
    np.testing.assert_allclose(df.values.astype(np.float32), emb[2].cpu().numpy(), rtol=0, atol=0)
