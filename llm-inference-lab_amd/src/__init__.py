"""Mirror of the reference's `src` package for the speculative-decoding hot path.

Only what the path needs is here: `src.kernels` (registry + HIP ops),
`src.specdec` (pipeline, policies, controllers, cache types, model wrappers) and
`src.scheduler`. Everything computes through the gfx950 C-ABI library.
"""
