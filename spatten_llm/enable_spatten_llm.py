from spatten_amd.enable_spatten_llm import enable_spatten_llm  # noqa: F401

__all__ = ["enable_spatten_llm"]
