from spatten_amd.kv_cache_token_pruning import DIM_TO_SLICE, SpAttenKVCache, slice1d, slice2d, slice3d  # noqa: F401
