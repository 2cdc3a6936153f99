"""A GraphSAGE-style node classifier trained in PyTorch on batches the glx engine samples on the GPU.

    python examples/train_sage_pytorch.py [epochs]            (needs one GPU)

What the reference's examples/pytorch/gcn does with its Dataset + PyG loader, on this engine: a GSL query
(seed batch -> 10 neighbours -> 5 neighbours of each) read through graphlearn.python.nn.pytorch.Dataset with
device="cuda" (the two hops run as ONE engine call; ids and float attributes never leave HBM) and prefetch=True
(batches sampled ahead of the optimiser step).  The model is plain torch: two mean-aggregation layers.

The graph is synthetic: 20,000 vertices in 5 classes, 16 noisy features whose mean depends weakly on the class, and
edges that stay inside the class 80 % of the time -- a vertex is hard to classify from its own features, easy from its
neighbourhood's, so the accuracy shows that the sampled neighbourhoods are the right ones.
"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd", "python"))

import torch  # noqa: E402
import graphlearn as gl  # noqa: E402
import graphlearn.python.nn.pytorch as thg  # noqa: E402  (the reference's import path)

V, CLASSES, DIM, DEG = 20000, 5, 16, 12
FANOUT = (10, 5)
BATCH = 512


def write_sources(directory):
    rng = np.random.default_rng(0)
    label = rng.integers(0, CLASSES, V)
    centers = rng.standard_normal((CLASSES, DIM)) * 0.35
    feats = centers[label] + rng.standard_normal((V, DIM))
    by_class = [np.flatnonzero(label == c) for c in range(CLASSES)]
    npath, epath = os.path.join(directory, "node"), os.path.join(directory, "edge")
    with open(npath, "w") as f:
        f.write("id:int64\tlabel:int64\tfeature:string\n")
        for v in range(V):
            f.write("%d\t%d\t%s\n" % (v, label[v], ":".join("%.4f" % x for x in feats[v])))
    with open(epath, "w") as f:
        f.write("src_id:int64\tdst_id:int64\n")
        for v in range(V):
            same = rng.random(DEG) < 0.8
            dst = np.where(same, rng.choice(by_class[label[v]], DEG), rng.integers(0, V, DEG))
            f.writelines("%d\t%d\n" % (v, d) for d in dst)
    return npath, epath


class Sage(torch.nn.Module):
    """h_v = relu(W [x_v || mean of its sampled neighbours' h]) twice, then a linear classifier."""

    def __init__(self, dim, hidden, classes):
        super().__init__()
        self.l1 = torch.nn.Linear(2 * dim, hidden)
        self.l2 = torch.nn.Linear(hidden + dim, hidden)
        self.out = torch.nn.Linear(hidden, classes)

    def forward(self, x0, x1, x2):
        # x0 [B, D], x1 [B * f1, D], x2 [B * f1 * f2, D]
        b, f1, f2 = x0.shape[0], FANOUT[0], FANOUT[1]
        h1 = torch.relu(self.l1(torch.cat([x1, x2.view(b * f1, f2, -1).mean(1)], dim=1)))   # hop-1 vertices
        h0 = torch.relu(self.l2(torch.cat([x0, h1.view(b, f1, -1).mean(1)], dim=1)))        # seeds
        return self.out(h0)


def main(epochs=3, quiet=False):
    d = tempfile.mkdtemp(prefix="glx_sage_")
    npath, epath = write_sources(d)
    gl.set_padding_mode(gl.CIRCULAR)
    gl.set_sampling_seed(7)
    g = gl.Graph() \
        .node(npath, "n", gl.Decoder(labeled=True, attr_types=["float"] * DIM)) \
        .edge(epath, ("n", "n", "e"), gl.Decoder()) \
        .init()
    query = g.V("n").batch(BATCH).shuffle(traverse=True).alias("seed") \
             .outV("e").sample(FANOUT[0]).by("random").alias("hop1") \
             .outV("e").sample(FANOUT[1]).by("random").alias("hop2") \
             .values()
    data = thg.Dataset(query, window=4, device="cuda", prefetch=True)
    model = Sage(DIM, 64, CLASSES).cuda()
    opt = torch.optim.Adam(model.parameters(), lr=5e-3)
    history = []
    for epoch in range(epochs):
        t0, seen, correct, loss_sum, batches = time.time(), 0, 0, 0.0, 0
        for batch in data:  # one epoch: every vertex once, in random order
            seed, hop1, hop2 = batch["seed"], batch["hop1"], batch["hop2"]
            logits = model(seed.float_attrs, hop1.float_attrs, hop2.float_attrs)
            labels = seed.labels.long()
            loss = torch.nn.functional.cross_entropy(logits, labels)
            opt.zero_grad()
            loss.backward()
            opt.step()
            seen += labels.shape[0]
            correct += int((logits.argmax(1) == labels).sum())
            loss_sum += float(loss.detach())
            batches += 1
        history.append((loss_sum / batches, correct / seen))
        if not quiet:
            print("epoch %d: loss %.3f, accuracy %.3f, %d vertices in %.2f s (%d sampled edges per batch)"
                  % (epoch, history[-1][0], history[-1][1], seen, time.time() - t0, BATCH * FANOUT[0] * (1 + FANOUT[1])))
    data.close()
    g.close()
    return history


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
