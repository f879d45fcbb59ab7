"""``spatten_llm.utils`` import path of the reference, re-exporting the MI355X package's helpers."""
from spatten_amd.utils import load_jsonl, load_mt_bench_prompts  # noqa: F401
