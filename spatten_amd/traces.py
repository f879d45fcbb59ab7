"""Reader for the reference's cascade-schedule traces (spatten_hardware/hardware/workloads/*.csv).

Schema (workloads/small.csv:1): one row per (iteration, layer, head) request of the accelerator model —
``iteration_id, layer_id, head_id, embedding_length_D, sentence_length_L, key_fetch_num, quant_key_bit,
quant_query_bit, auto_requant_thres, if_requant, auto_requant_incre, value_fetch_num, quant_value_bit,
if_accumulate_importance, if_rescale_previous_importance, if_topk, topk``.

The traces carry no tensors; what they give is a realistic cascade SCHEDULE: how many keys each layer
still fetches (global token pruning), how many values (local V pruning), which heads survive (head pruning shows as
missing head rows), and the progressive-quantisation threshold.  ``CascadeSchedule.fractions()`` turns a trace into
per-layer keep ratios that ``run_spatten_synthetic.py --trace`` applies to a model of any size, and
``CascadeSchedule.pq_profile()`` the bit profile (key MSB bits, value bits) the accelerator model runs the trace at —
the mapping of the reference's harness (TestSpAtten.scala:64-97): ``quant_key_bit`` -1 / 10 / 12 -> 8 with the requant
forced on, ``quant_value_bit`` -1 / 10 / 12 -> 8; 6 is the fused (6, 2) fetch profile (TestSpAtten.scala:173-176);
``auto_requant_incre`` = the LSB bits added by a refetch (4 in every trace, the RTL's requantBitCount,
SpAttenController.scala:35).  ``if_rescale_previous_importance`` is carried as a field; NO code of the reference reads it
(it is not among the columns TestSpAtten.scala maps, and the Python plugin has no accumulated importance at all), so there
is no rule to restate — ``LayerStep.rescale_previous_importance`` only reports what the trace says.
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
    key_bits: int = -1         # quant_key_bit as written in the trace (-1: unquantised in the software model)
    value_bits: int = -1       # quant_value_bit
    lsb_bits: int = -1         # auto_requant_incre: bits a refetch adds
    rescale_previous_importance: bool = False
    requant: bool = False
    requant_threshold: float = -1.0
    accumulate_importance: bool = False
    next_keys: int = -1        # `topk` of the row flagged if_topk: the token set the NEXT layer starts from
    keys_uniform: bool = True  # every head row of the layer carries the same key_fetch_num (ONE token set for all heads)


@dataclass
class CascadeSchedule:
    iterations: Dict[int, List[LayerStep]]

    def layers(self, iteration: int = 0) -> List[LayerStep]:
        return self.iterations[iteration]

    def token_scope(self, iteration: int = 0) -> str:
        """"global" when every layer of the iteration carries ONE ``key_fetch_num`` for all of its heads — the accelerator
        model's token pruning, one kept set per layer (workloads/small.csv:1) — else "head".  What
        ``enable_spatten_llm(..., token_scope=...)`` should be given for this trace."""
        return "global" if all(s.keys_uniform for s in self.layers(iteration)) else "head"

    def pq_profile(self, iteration: int = 0):
        """(key MSB bits, value bits) the accelerator model fetches this trace at — TestSpAtten.scala:64-97: a key width of
        -1 / 10 / 12 runs as 8 bits with the requant on, a value width of -1 / 10 / 12 as 8 (a layer written -1 / -1 inside a
        quantised trace therefore counts as (8, 8)) — or None when NO layer of the iteration carries a quantisation width at
        all (the software model's unquantised traces).  The widest profile over the layers (they agree in the reference's
        traces)."""
        best = None
        steps = self.layers(iteration)
        if all(s.key_bits == -1 and s.value_bits == -1 for s in steps):
            return None
        for s in steps:
            kb = 8 if s.key_bits in (-1, 10, 12) else s.key_bits
            vb = 8 if s.value_bits in (-1, 10, 12) else s.value_bits
            best = (kb, vb) if best is None else (max(best[0], kb), max(best[1], vb))
        return best

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
        header = next(rd, None)
        if header is None:
            raise ValueError(f"{path}: empty file, not a SpAtten workload trace")
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
            if len(st.heads) > 1 and st.keys != int(r["key_fetch_num"]):
                st.keys_uniform = False
            st.keys = int(r["key_fetch_num"])
            st.values = int(r["value_fetch_num"])
            st.key_bits, st.value_bits = int(r["quant_key_bit"]), int(r["quant_value_bit"])
            st.lsb_bits = int(r["auto_requant_incre"])
            st.rescale_previous_importance = st.rescale_previous_importance or _b(r["if_rescale_previous_importance"])
            st.requant = st.requant or _b(r["if_requant"])
            st.requant_threshold = float(r["auto_requant_thres"])
            st.accumulate_importance = _b(r["if_accumulate_importance"])
            if _b(r["if_topk"]):
                st.next_keys = int(r["topk"])
    return CascadeSchedule({it: [layers[k] for k in sorted(layers)] for it, layers in sorted(iters.items())})
