"""Small host utilities of the pipeline (interfaces, token validation, determinism)."""
from .deterministic import ensure_deterministic, set_deterministic_mode  # noqa: F401
from .interfaces import LanguageModel  # noqa: F401
from .token_validation import get_vocab_size, validate_and_clamp_tokens  # noqa: F401
