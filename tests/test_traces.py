"""Trace-CSV reader (schema of the reference's workloads/*.csv) on a self-made trace.  CPU only."""
import os

import pytest

from spatten_amd.traces import read_trace

HERE = os.path.dirname(os.path.abspath(__file__))


def test_read_synthetic_trace():
    s = read_trace(os.path.join(HERE, "golden", "trace_synthetic.csv"))
    steps = s.layers(0)
    assert [st.layer for st in steps] == [0, 1, 2]
    assert [len(st.heads) for st in steps] == [4, 4, 3]
    assert steps[0].next_keys == 600 and steps[1].next_keys == 300 and steps[2].next_keys == -1
    fr = s.fractions(0)
    assert [round(f["token_keep"], 2) for f in fr] == [1.0, 0.6, 0.3]
    assert all(abs(f["value_keep"] - 0.3) < 1e-9 for f in fr)
    assert [f["head_keep"] for f in fr] == [1.0, 1.0, 0.75]
    assert fr[0]["requant_threshold"] == 0.05 and steps[0].accumulate_importance
    # the bit columns: 6-bit keys + 4 LSBs on refetch, 6-bit values — the (6, 2) fused profile of the per8 trace
    assert (steps[0].key_bits, steps[0].lsb_bits, steps[0].value_bits) == (6, 4, 6) and s.pq_profile(0) == (6, 6)
    assert not steps[0].rescale_previous_importance
    # one key_fetch_num per layer for all heads: the accelerator model's GLOBAL token pruning (one kept set per layer)
    assert all(st.keys_uniform for st in steps) and s.token_scope(0) == "global"


def test_pq_profile_follows_the_harness_mapping(tmp_path):
    """TestSpAtten.scala:64-97: key width -1 / 10 / 12 runs as 8 bits (requant on), value width -1 / 10 / 12 as 8."""
    from spatten_amd.traces import COLUMNS
    rows = [",".join(COLUMNS),
            "0,0,0,64.0,100,100,12,16,-1,False,-1,100,8,True,True,False,-1",
            "0,1,0,64.0,100,80,-1,-1,-1,False,-1,80,-1,True,False,False,-1"]
    p = tmp_path / "t.csv"
    p.write_text("\n".join(rows) + "\n")
    s = read_trace(str(p))
    assert s.pq_profile(0) == (8, 8)
    assert s.layers(0)[0].rescale_previous_importance and not s.layers(0)[1].rescale_previous_importance
    # a layer written -1 / -1 inside a quantised trace counts as (8, 8) — here layer 0 is (6, 6), layer 1 is -1 / -1
    rows2 = [rows[0], "0,0,0,64.0,100,100,6,16,-1,False,-1,100,6,True,True,False,-1",
             "0,0,1,64.0,100,90,6,16,-1,False,-1,100,6,True,True,False,-1", rows[2]]
    p2 = tmp_path / "t2.csv"
    p2.write_text("\n".join(rows2) + "\n")
    s2 = read_trace(str(p2))
    assert s2.pq_profile(0) == (8, 8)
    assert not s2.layers(0)[0].keys_uniform and s2.token_scope(0) == "head"      # heads of layer 0 fetch 100 and 90 keys
    # a trace without any quantisation width: no profile
    p3 = tmp_path / "t3.csv"
    p3.write_text("\n".join([rows[0], rows[2]]) + "\n")
    assert read_trace(str(p3)).pq_profile(0) is None


def test_rejects_other_csv(tmp_path):
    p = tmp_path / "x.csv"
    p.write_text("a,b,c\n1,2,3\n")
    with pytest.raises(ValueError):
        read_trace(str(p))


def test_an_empty_trace_file_is_refused_with_a_clear_error(tmp_path):
    import pytest
    from spatten_amd import traces
    f = tmp_path / "empty.csv"
    f.write_text("")
    with pytest.raises(ValueError, match="empty"):
        traces.read_trace(str(f))
