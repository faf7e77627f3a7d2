"""Synthetic inputs of the parity tests: the neutral module rawaudiovae_kelsey_amd/synth.py (bench.py uses it
directly), re-exported here under the name the tests and the oracle have always imported."""
from rawaudiovae_kelsey_amd.synth import (PARAM_NAMES, flops_per_frame, make_eps, make_frames, make_params,  # noqa: F401
                                          num_params, param_shapes)
