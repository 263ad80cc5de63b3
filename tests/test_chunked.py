"""Chunked arguments through CallFunction (compute.ChunkedDatum).  Values are checked against Arrow C++ on the
logical arrays; the CHUNK LAYOUT of every result against the reference's rules — spans end where any
argument's chunk ends (iterateExecSpans, arrow/compute/executor.go:750-870), empty outputs are dropped
(WrapResults :521-582, :1000-1080), `take` follows selection.go:195-330, unique / sort_indices return one
array, cumulative_sum one chunk (vector_cumulative.go:368-391)."""
import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pytest

from arrow_go_amd import compute as ac

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sess():
    s = ac.Session(0)
    yield s
    s.close()


def chunked(arr, cuts):
    """split `arr` at the given positions (may repeat → empty chunks)"""
    edges = [0] + list(cuts) + [len(arr)]
    return pa.chunked_array([arr.slice(a, b - a) for a, b in zip(edges[:-1], edges[1:])], type=arr.type)


def span_lengths(n, *cut_lists):
    edges = sorted(set([0, n] + [c for cl in cut_lists for c in cl]))
    return [b - a for a, b in zip(edges[:-1], edges[1:]) if b > a]


def layout(c):
    return [len(x) for x in c.chunks]


def rand(rng, typ, n, p_null=0.1):
    mask = rng.random(n) < p_null
    if pa.types.is_boolean(typ):
        return pa.array(rng.random(n) < 0.5, mask=mask, type=typ)
    if pa.types.is_string(typ) or pa.types.is_large_string(typ):
        words = ["", "a", "bc", "hello world", "x" * 30] + ["w%d" % i for i in range(40)]
        return pa.array([words[j] for j in rng.integers(0, len(words), n)], mask=mask, type=typ)
    if pa.types.is_floating(typ):
        return pa.array(rng.integers(-1000, 1000, n).astype(np.float64), mask=mask, type=typ)
    lo = 0 if pa.types.is_unsigned_integer(typ) else -100
    return pa.array(rng.integers(lo, 100, n), mask=mask, type=typ)


def test_scalar_functions_follow_span_boundaries(sess):
    rng = np.random.default_rng(1)
    n = 10007
    a, b = rand(rng, pa.int64(), n), rand(rng, pa.int64(), n)
    ca, cb = chunked(a, [100, 100, 5000, 9000]), chunked(b, [64, 5000, 7777])
    out = sess.call_function("add", [ca, cb])
    assert isinstance(out, pa.ChunkedArray) and layout(out) == span_lengths(n, [100, 5000, 9000], [64, 5000, 7777])
    assert out.combine_chunks().equals(pc.add_checked(a, b))
    out = sess.call_function("greater", [ca, b])           # chunked ∘ array
    assert layout(out) == span_lengths(n, [100, 5000, 9000]) and out.combine_chunks().equals(pc.greater(a, b))
    out = sess.call_function("multiply_unchecked", [cb, pa.scalar(3, pa.int64())])   # chunked ∘ scalar
    assert layout(out) == span_lengths(n, [64, 5000, 7777]) and out.combine_chunks().equals(pc.multiply(b, 3))
    out = sess.call_function("cast", [ca], "to_type=double")
    assert layout(out) == span_lengths(n, [100, 5000, 9000]) and out.combine_chunks().equals(a.cast(pa.float64()))
    out = sess.call_function("is_in", [ca], value_set=pa.array([1, 2, 3, None], type=pa.int64()))
    assert out.combine_chunks().equals(pc.is_in(a, value_set=pa.array([1, 2, 3, None], type=pa.int64())))
    x, y = rand(rng, pa.bool_(), n), rand(rng, pa.bool_(), n)
    out = sess.call_function("and_kleene", [chunked(x, [13, 4096]), chunked(y, [8191])])
    assert layout(out) == span_lengths(n, [13, 4096], [8191]) and out.combine_chunks().equals(pc.and_kleene(x, y))
    # mismatched total lengths
    with pytest.raises(ac.ErrInvalid, match="same length"):
        sess.call_function("add", [ca, chunked(b.slice(0, n - 1), [10])])
    # no chunks at all / only empty chunks: one empty chunk comes back (the first output is kept)
    out = sess.call_function("add", [pa.chunked_array([], type=pa.int64()), pa.chunked_array([], type=pa.int64())])
    assert isinstance(out, pa.ChunkedArray) and len(out) == 0 and out.type == pa.int64()


@pytest.mark.parametrize("typ", [pa.int64(), pa.int16(), pa.float64(), pa.bool_(), pa.string(), pa.large_string()], ids=str)
def test_filter_and_take_chunked(sess, typ):
    rng = np.random.default_rng(2)
    n = 6001
    v, m = rand(rng, typ, n), rand(rng, pa.bool_(), n)
    vc, mc = [50, 3000, 3000], [2048, 4096]
    for opt, null_sel in (("", "drop"), ("null_selection_behavior=emit_null", "emit_null")):
        out = sess.call_function("filter", [chunked(v, vc), chunked(m, mc)], opt)
        exp = pc.filter(v, m, null_selection_behavior=null_sel)
        assert out.combine_chunks().equals(exp)
        edges = sorted(set([0, n] + vc + mc))
        sel = [len(pc.filter(v.slice(a, b - a), m.slice(a, b - a), null_selection_behavior=null_sel)) for a, b in zip(edges[:-1], edges[1:])]
        assert layout(out) == [k for k in sel if k > 0]
    out = sess.call_function("filter", [v, chunked(m, mc)])      # array values, chunked mask
    assert out.combine_chunks().equals(pc.filter(v, m)) and isinstance(out, pa.ChunkedArray)
    idx = pa.array(rng.integers(0, n, 2500), mask=rng.random(2500) < 0.1, type=pa.int32())
    out = sess.call_function("take", [chunked(v, vc), idx])      # chunked values, array indices → one chunk
    assert layout(out) == [2500] and out.combine_chunks().equals(pc.take(v, idx))
    out = sess.call_function("take", [v, chunked(idx, [700, 700, 1999])])   # array values, chunked indices → chunk per indices chunk
    assert layout(out) == [700, 1299, 501] and out.combine_chunks().equals(pc.take(v, idx))
    out = sess.call_function("take", [chunked(v, vc), chunked(idx, [1, 2000])])
    assert layout(out) == [1, 1999, 500] and out.combine_chunks().equals(pc.take(v, idx))
    with pytest.raises(ac.ErrIndex):
        sess.call_function("take", [chunked(v, vc), pa.array([0, n], type=pa.int64())])   # bounds are those of the whole column


@pytest.mark.parametrize("typ", [pa.int64(), pa.uint8(), pa.float64(), pa.string()], ids=str)
def test_hash_kernels_chunked(sess, typ):
    rng = np.random.default_rng(3)
    n = 9000
    v = rand(rng, typ, n)
    c = chunked(v, [1, 1, 4500, 8999])
    out = sess.call_function("unique", [c])                      # vector_hash_test.go:420-449 TestUniqueChunkedArrayInvoke
    assert isinstance(out, pa.Array) and out.equals(pc.unique(v))
    for opt, enc in (("", "mask"), ("null_encoding_behavior=encode", "encode")):
        out = sess.call_function("dictionary_encode", [c], opt)  # :925-… TestDictionaryEncodeChunked*: one dictionary for all chunks
        exp = pc.dictionary_encode(v, null_encoding=enc)
        assert layout(out) == [1, 4499, 4499, 1]
        for ch in out.chunks:
            assert ch.dictionary.equals(exp.dictionary)
        assert pa.concat_arrays([ch.indices for ch in out.chunks]).equals(exp.indices)
    if pa.types.is_string(typ):
        a1, a2 = pa.array(["foo", "bar", "foo"]), pa.array(["bar", "baz", "quuux", "foo"])
        assert sess.call_function("unique", [pa.chunked_array([a1, a2])]).to_pylist() == ["foo", "bar", "baz", "quuux"]


def test_cumulative_sum_and_sort_chunked(sess):
    rng = np.random.default_rng(4)
    n = 20011
    v = rand(rng, pa.int64(), n)
    c = chunked(v, [3, 10000, 10000, 15000])
    out = sess.call_function("cumulative_sum", [c])
    assert layout(out) == [n] and out.combine_chunks().equals(pc.cumulative_sum(v))
    out = sess.call_function("cumulative_sum", [c], "skip_nulls=1;start=int64:5")
    assert out.combine_chunks().equals(pc.cumulative_sum(v, start=5, skip_nulls=True))
    idx = sess.call_function("sort_indices", [c], "order=descending;null_placement=at_start")
    assert isinstance(idx, pa.Array) and idx.equals(pc.sort_indices(v, sort_keys=[("", "descending")], null_placement="at_start"))
    out = sess.call_function("sort", [c], "order=ascending")
    assert isinstance(out, pa.ChunkedArray) and layout(out) == [n]
    assert out.combine_chunks().equals(pc.take(v, pc.sort_indices(v)))
    f = rand(rng, pa.float64(), n)
    out = sess.call_function("cumulative_sum", [chunked(f, [7777])])
    assert out.combine_chunks().equals(pc.cumulative_sum(f))     # integer-valued doubles: exact in any order


def test_chunked_results_feed_the_next_call(sess):
    rng = np.random.default_rng(5)
    a = rand(rng, pa.int64(), 5000, 0.0)
    c = chunked(a, [1234, 4000])
    s1 = sess.call_function("add", [c, c], keep_on_device=True)
    s2 = sess.call_function("greater", [s1, pa.scalar(0, pa.int64())], keep_on_device=True)
    out = sess.call_function("filter", [s1, s2])
    exp = pc.filter(pc.add(a, a), pc.greater(pc.add(a, a), 0))
    assert out.combine_chunks().equals(exp)


# ---- record batches (compute.RecordDatum): FilterRecordBatch selection.go:679-722, takeRecordImpl :160-204 ----------
def test_record_batch_filter_take_sort(sess):
    rng = np.random.default_rng(6)
    n = 7001
    rb = pa.RecordBatch.from_arrays([rand(rng, pa.int64(), n), rand(rng, pa.float64(), n), rand(rng, pa.string(), n), rand(rng, pa.bool_(), n),
                                     rand(rng, pa.int8(), n, 0.0)], names=["a", "b", "s", "t", "z"])
    m = rand(rng, pa.bool_(), n)
    for opt, null_sel in (("", "drop"), ("null_selection_behavior=emit_null", "emit_null")):
        out = sess.call_function("filter", [rb, m], opt)
        assert isinstance(out, pa.RecordBatch) and out.schema.names == rb.schema.names
        assert out.equals(rb.filter(m, null_selection_behavior=null_sel))
    with pytest.raises(ac.ErrInvalid, match="same length"):
        sess.call_function("filter", [rb, m.slice(1)])
    with pytest.raises(ac.ErrNotImplemented, match="only implemented for Array filter"):
        sess.call_function("filter", [rb, chunked(m, [10])])
    idx = pa.array(rng.integers(0, n, 3000), mask=rng.random(3000) < 0.1, type=pa.int32())
    assert sess.call_function("take", [rb, idx]).equals(rb.take(idx))
    assert sess.call_function("take", [rb, chunked(idx, [5, 1000])]).equals(rb.take(idx))
    with pytest.raises(ac.ErrIndex):
        sess.call_function("take", [rb, pa.array([n], type=pa.int64())])
    keys = "sort_keys=4:asc:at_end,0:desc:at_start"
    exp = pc.sort_indices(rb, sort_keys=[("z", "ascending"), ("a", "descending")], null_placement="at_start")
    got = sess.call_function("sort_indices", [rb], keys)
    # z has no nulls, so only key a's placement (at_start) matters — same reading in both libraries
    assert got.equals(exp)
    with pytest.raises(ac.ErrNotImplemented, match="record batch"):
        sess.call_function("add", [rb, rb])


def test_chunked_dictionary_arrays_with_different_dictionaries(sess):
    """array.Concatenate over dictionary chunks unifies the dictionaries and transposes the indices
    (arrow/array/concat.go:600-640, dictionary.go:1380-1500) — here on the device; values against Arrow C++"""
    rng = np.random.default_rng(9)
    d1 = pa.DictionaryArray.from_arrays(pa.array(rng.integers(0, 4, 900), mask=rng.random(900) < 0.1, type=pa.int32()), pa.array(["a", "b", "c", "d"]))
    d2 = pa.DictionaryArray.from_arrays(pa.array(rng.integers(0, 5, 1100), mask=rng.random(1100) < 0.1, type=pa.int32()), pa.array(["c", "x", None, "a", "y"]))
    d3 = pa.DictionaryArray.from_arrays(pa.array(rng.integers(0, 2, 300), type=pa.int32()), pa.array(["y", "a"]))
    c = pa.chunked_array([d1, d2, d3])
    ref = pa.concat_arrays([x.cast(pa.string()) for x in (d1, d2, d3)])   # the logical column (Arrow C++ cannot unify dictionaries holding nulls)
    logical = lambda arr: (arr if isinstance(arr, pa.Array) else arr.combine_chunks()).cast(pa.string()).to_pylist()
    sel = pa.array(rng.integers(0, len(c), 500), mask=rng.random(500) < 0.1, type=pa.int64())
    got = sess.call_function("take", [c, sel])
    assert logical(got) == logical(pc.take(ref, sel))
    # the unified dictionary is the unifier's: values in first-seen order over the chunk dictionaries, null included
    assert got.chunk(0).dictionary.to_pylist() == ["a", "b", "c", "d", "x", None, "y"]
    mask = pa.array(rng.random(len(c)) < 0.4)
    assert logical(sess.call_function("filter", [c, mask])) == logical(pc.filter(ref, mask))
    # unique hashes the (transposed) INDICES (dictionaryHashState, vector_hash.go:505-576): a null index and an index of
    # the dictionary's null entry are two different results
    uq = sess.call_function("unique", [c])
    unified = ["a", "b", "c", "d", "x", None, "y"]
    moved = []
    for ch in (d1, d2, d3):
        vals = ch.dictionary.to_pylist()
        moved += [None if i is None else unified.index(vals[i]) for i in ch.indices.to_pylist()]
    first_seen = list(dict.fromkeys(moved))
    assert uq.indices.to_pylist() == first_seen and uq.dictionary.to_pylist() == unified
    # numeric values
    n1 = pa.DictionaryArray.from_arrays(pa.array([0, 1, 1, None], pa.int32()), pa.array([10, 20], pa.int64()))
    n2 = pa.DictionaryArray.from_arrays(pa.array([1, 0, 2], pa.int32()), pa.array([20, 30, 10], pa.int64()))
    got = sess.call_function("take", [pa.chunked_array([n1, n2]), pa.array([6, 0, 3, 4], pa.int32())])
    assert got.combine_chunks().cast(pa.int64()).to_pylist() == [10, 10, None, 30]
    assert got.chunk(0).dictionary.to_pylist() == [10, 20, 30]
