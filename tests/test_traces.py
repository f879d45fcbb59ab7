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


def test_rejects_other_csv(tmp_path):
    p = tmp_path / "x.csv"
    p.write_text("a,b,c\n1,2,3\n")
    with pytest.raises(ValueError):
        read_trace(str(p))
