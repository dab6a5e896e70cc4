"""Operators: fused learner ops, flat optimizer, HIP inference engine, HIP self-play engine."""
