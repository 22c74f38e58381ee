"""Model wrappers (reference: src/specdec/models/)."""
from .hip_lm import HipLM, IdTokenizer, create_hip_lm  # noqa: F401
from .fake_lm import FakeLM, create_fake_lm  # noqa: F401
