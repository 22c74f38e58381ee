"""KV cache containers (reference: src/specdec/cache/)."""
from .kv_types import KVCache, validate_kv_compatibility  # noqa: F401
