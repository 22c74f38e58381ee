"""Acceptance policies and K controllers (reference: src/specdec/policies/)."""
from .controllers import AdaptiveKController, FixedKController, KController, create_controller  # noqa: F401
from .policies import (  # noqa: F401
    AcceptancePolicy, ConfidenceThresholdPolicy, LongestPrefixPolicy, TopKAgreementPolicy,
    TypicalAcceptancePolicy, create_policy,
)
