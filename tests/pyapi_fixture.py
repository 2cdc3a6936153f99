"""Test data for the Python-API tests: the synthetic TSV sources and the closed-form
expectations of the reference's own Python unit tests (graphlearn/python/tests/utils.py:
gen_node_data / gen_edge_data / fixed_dst_ids and the check_* helpers), restated.

Node v of type T:   weight v/10, label v, attributes "v:float(v):v:hehe".
Edge s -> d:        weight (s + 0.1 d)/10, label s, attributes "s:float(s):d:hehe",
                    d in { s*i % (hi - lo) + lo : i = 1 .. s % 5 }  (so s % 5 == 0 has none).
With attr_types ['int', 'float', 'string', ('string', 10)] a record stores
ints [v, Hash64('hehe') % 10], floats [float(v)], strings [third field].
"""
import os

import numpy as np

WEIGHTED, LABELED, ATTRIBUTED = "weighted", "labeled", "attributed"
ATTR_TYPES = ["int", "float", "string", ("string", 10)]


def fixed_dst_ids(src, dst_range):
    lo, hi = dst_range
    srcs = [src] if np.isscalar(src) else list(src)
    return [int(s) * i % (hi - lo) + lo for s in srcs for i in range(1, int(s) % 5 + 1)]


def _header(id_columns, schema):
    cols = list(id_columns)
    if WEIGHTED in schema:
        cols.append("weight:float")
    if LABELED in schema:
        cols.append("label:int64")
    if ATTRIBUTED in schema:
        cols.append("feature:string")
    return "\t".join(cols) + "\n"


def write_nodes(directory, name, id_range, schema):
    path = os.path.join(directory, name)
    with open(path, "w") as f:
        f.write(_header(["id:int64"], schema))
        for v in range(*id_range):
            rec = ["%d" % v]
            if WEIGHTED in schema:
                rec.append("%f" % (v / 10.0))
            if LABELED in schema:
                rec.append("%d" % v)
            if ATTRIBUTED in schema:
                rec.append("%d:%f:%d:%s" % (v, float(v), v, "hehe"))
            f.write("\t".join(rec) + "\n")
    return path


def write_edges(directory, name, src_range, dst_range, schema):
    path = os.path.join(directory, name)
    with open(path, "w") as f:
        f.write(_header(["src_id:int64", "dst_id:int64"], schema))
        for s in range(*src_range):
            for d in fixed_dst_ids(s, dst_range):
                rec = ["%d" % s, "%d" % d]
                if WEIGHTED in schema:
                    rec.append("%f" % ((s + 0.1 * d) / 10.0))
                if LABELED in schema:
                    rec.append("%d" % s)
                if ATTRIBUTED in schema:
                    rec.append("%d:%f:%d:%s" % (s, float(s), d, "hehe"))
                f.write("\t".join(rec) + "\n")
    return path


def write_entity_nodes(directory, name, count=120):
    """id, label = id, four float attributes 0.1 id .. 0.4 id (utils.gen_entity_node)."""
    path = os.path.join(directory, name)
    with open(path, "w") as f:
        f.write("id:int64\tlabel:int64\tfeature:string\n")
        for v in range(count):
            f.write("%d\t%d\t%f:%f:%f:%f\n" % (v, v, v * 0.1, v * 0.2, v * 0.3, v * 0.4))
    return path


def write_relation_edges(directory, name, count=100):
    """i -> i+2, i+3, i+5, each with weight i/100 (utils.gen_relation_edge)."""
    path = os.path.join(directory, name)
    with open(path, "w") as f:
        f.write("src_id:int64\tdst_id:int64\tweight:float\n")
        for i in range(count):
            for step in (2, 3, 5):
                f.write("%d\t%d\t%f\n" % (i, i + step, i / 100.0))
    return path


def write_cond_nodes(directory, name, count=200):
    """id, weight 0.1 id, attributes "id % 5 : id % 4 : 0.3 id : str(id % 3)" (utils.gen_cond_node; attr types
    COND_ATTR_TYPES = ['int', 'int', 'float', 'string'])."""
    path = os.path.join(directory, name)
    with open(path, "w") as f:
        f.write("id:int64\tweight:float\tfeature:string\n")
        for i in range(count):
            f.write("%d\t%f\t%d:%d:%f:%s\n" % (i, i * 0.1, i % 5, i % 4, i * 0.3, str(i % 3)))
    return path


# ---- expectations -------------------------------------------------------------------
def expect_edges_follow_generator(edges, dst_range, seed_ids, default_dst_id):
    src = edges.src_ids.reshape(-1)
    dst = edges.dst_ids.reshape(-1)
    assert set(src.tolist()) <= set(int(x) for x in seed_ids)
    for s, d in zip(src.tolist(), dst.tolist()):
        if s % 5 == 0:
            assert d == default_dst_id, (s, d)
        else:
            assert d in fixed_dst_ids(s, dst_range), (s, d)


def expect_edge_columns(edges, weighted=False, labeled=False, attributed=False):
    """Columns of existing edges are the generator's closed forms."""
    src, dst = edges.src_ids, edges.dst_ids
    if weighted:
        np.testing.assert_almost_equal(edges.weights, 0.1 * (src + 0.1 * dst), decimal=5)
    if labeled:
        np.testing.assert_equal(edges.labels, src)
    if attributed:
        np.testing.assert_equal(edges.int_attrs[..., 0], src)
        np.testing.assert_almost_equal(edges.float_attrs[..., 0], src.astype(np.float64), decimal=5)
        np.testing.assert_equal(edges.string_attrs[..., 0], np.vectorize(str)(dst))


def expect_default_edge_columns(edges, labeled=False, attributed=False, default_int=0, default_float=0.0,
                                default_string=""):
    n = int(np.prod(edges.shape))
    if labeled:
        np.testing.assert_equal(edges.labels.reshape(-1), [-1] * n)
    if attributed:
        np.testing.assert_equal(edges.int_attrs.reshape(-1), [default_int] * (2 * n))
        np.testing.assert_almost_equal(edges.float_attrs.reshape(-1), [default_float] * n, decimal=4)
        np.testing.assert_equal(edges.string_attrs.reshape(-1), [default_string] * n)


def expect_node_columns(nodes, weighted=False, labeled=False, attributed=False):
    ids = nodes.ids
    if weighted:
        np.testing.assert_almost_equal(nodes.weights, 0.1 * ids, decimal=5)
    if labeled:
        np.testing.assert_equal(nodes.labels, ids)
    if attributed:
        np.testing.assert_equal(nodes.int_attrs[..., 0], ids)
        np.testing.assert_almost_equal(nodes.float_attrs[..., 0], ids.astype(np.float64), decimal=5)
        np.testing.assert_equal(nodes.string_attrs[..., 0], np.vectorize(str)(ids))


def expected_topk(seed_ids, dst_range, k, default_dst_id, padding_mode):
    """utils.check_topk_edge_ids: weights grow with dst, so top-k = largest dst ids first."""
    out = []
    for s in seed_ids:
        s = int(s)
        if s % 5 == 0:
            out.extend([default_dst_id] * k)
            continue
        best = sorted(fixed_dst_ids(s, dst_range), reverse=True)
        have = min(len(best), k)
        if padding_mode == "replicate":
            out.extend(best[:have] + [default_dst_id] * (k - have))
        else:
            out.extend((best[:have] * (k // have + 1))[:k])
    return out
