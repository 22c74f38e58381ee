"""Pipeline core (reference: src/specdec/core/)."""
