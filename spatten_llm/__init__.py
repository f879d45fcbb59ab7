"""Drop-in import path: ``spatten_llm`` (the reference's package name) re-exporting the MI355X
implementation in ``spatten_amd``.  ``from spatten_llm.enable_spatten_llm import enable_spatten_llm``
keeps working unchanged."""
from spatten_amd import SpAttenKVCache, enable_spatten_llm  # noqa: F401
