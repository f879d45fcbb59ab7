"""Caller-side helpers of the multi-turn harness — counterpart of the reference's ``spatten_llm/utils.py``
(the model / tokenizer loading and the MT-Bench download of that file need network and weights; what the hot path's
caller protocol consumes is the prompt list)."""
import json
from typing import List


def load_jsonl(file_path) -> List[dict]:
    """One JSON document per line -> list of dicts (spatten_llm/utils.py:105-112)."""
    list_data_dict = []
    with open(file_path, "r") as f:
        for line in f:
            if line.strip():
                list_data_dict.append(json.loads(line))
    return list_data_dict


def load_mt_bench_prompts(file_path) -> List[str]:
    """MT-Bench ``question.jsonl`` -> the flat list of turns the reference's driver iterates over
    (run_spatten_llama.py:104-107: ``prompts += sample["turns"]``)."""
    prompts: List[str] = []
    for sample in load_jsonl(file_path):
        prompts += sample["turns"]
    return prompts
