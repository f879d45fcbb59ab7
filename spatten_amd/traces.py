"""Reader for the reference's cascade-schedule traces (spatten_hardware/hardware/workloads/*.csv).

Schema (workloads/small.csv:1): one row per (iteration, layer, head) request of the accelerator model —
``iteration_id, layer_id, head_id, embedding_length_D, sentence_length_L, key_fetch_num, quant_key_bit,
quant_query_bit, auto_requant_thres, if_requant, auto_requant_incre, value_fetch_num, quant_value_bit,
if_accumulate_importance, if_rescale_previous_importance, if_topk, topk``.

The traces carry no tensors; what they give is a realistic cascade SCHEDULE: how many keys each layer
still fetches (global token pruning), how many values (local V pruning), which heads survive (head pruning shows as
missing head rows), and the progressive-quantisation threshold.  ``CascadeSchedule.fractions()`` turns a trace into
per-layer keep ratios that ``run_spatten_synthetic.py --trace`` applies to a model of any size.
"""
from __future__ import annotations

import csv
from dataclasses import dataclass, field
from typing import Dict, List

COLUMNS = ["iteration_id", "layer_id", "head_id", "embedding_length_D", "sentence_length_L", "key_fetch_num",
           "quant_key_bit", "quant_query_bit", "auto_requant_thres", "if_requant", "auto_requant_incre",
           "value_fetch_num", "quant_value_bit", "if_accumulate_importance", "if_rescale_previous_importance",
           "if_topk", "topk"]


def _b(x: str) -> bool:
    return x.strip().lower() == "true"


@dataclass
class LayerStep:
    layer: int
    heads: List[int] = field(default_factory=list)
    head_dim: int = 0
    length: int = 0            # sentence_length_L
    keys: int = 0              # key_fetch_num   (tokens still fetched by this layer)
    values: int = 0            # value_fetch_num (local V pruning)
    key_bits: int = -1
    value_bits: int = -1
    requant: bool = False
    requant_threshold: float = -1.0
    accumulate_importance: bool = False
    next_keys: int = -1        # `topk` of the row flagged if_topk: the token set the NEXT layer starts from


@dataclass
class CascadeSchedule:
    iterations: Dict[int, List[LayerStep]]

    def layers(self, iteration: int = 0) -> List[LayerStep]:
        return self.iterations[iteration]

    def fractions(self, iteration: int = 0):
        """Per layer: (token keep ratio, local-V keep ratio, head keep ratio, requant threshold or None)."""
        steps = self.layers(iteration)
        max_heads = max(len(s.heads) for s in steps)
        out = []
        for s in steps:
            L = max(s.length, 1)
            out.append({"layer": s.layer, "token_keep": s.keys / L, "value_keep": (s.values / s.keys) if s.keys else 0.0,
                        "head_keep": len(s.heads) / max_heads,
                        "requant_threshold": s.requant_threshold if s.requant else None})
        return out


def read_trace(path: str) -> CascadeSchedule:
    iters: Dict[int, Dict[int, LayerStep]] = {}
    with open(path, newline="") as f:
        rd = csv.reader(f)
        header = next(rd)
        if [h.strip() for h in header] != COLUMNS:
            raise ValueError(f"{path}: not a SpAtten workload trace (header {header[:3]}...)")
        for row in rd:
            if len(row) != len(COLUMNS) or not row[0].strip().lstrip("-").isdigit():
                continue                                   # trailing provenance lines (e.g. 'configs/...,,,')
            r = dict(zip(COLUMNS, row))
            it, layer, head = int(r["iteration_id"]), int(r["layer_id"]), int(r["head_id"])
            st = iters.setdefault(it, {}).setdefault(layer, LayerStep(layer=layer))
            st.heads.append(head)
            st.head_dim = int(float(r["embedding_length_D"]))
            st.length = int(r["sentence_length_L"])
            st.keys = int(r["key_fetch_num"])
            st.values = int(r["value_fetch_num"])
            st.key_bits, st.value_bits = int(r["quant_key_bit"]), int(r["quant_value_bit"])
            st.requant = st.requant or _b(r["if_requant"])
            st.requant_threshold = float(r["auto_requant_thres"])
            st.accumulate_importance = _b(r["if_accumulate_importance"])
            if _b(r["if_topk"]):
                st.next_keys = int(r["topk"])
    return CascadeSchedule({it: [layers[k] for k in sorted(layers)] for it, layers in sorted(iters.items())})
