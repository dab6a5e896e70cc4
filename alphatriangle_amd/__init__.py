"""alphatriangle_amd: MI355X-native AlphaZero self-play framework for the triangle puzzle."""

__version__ = "0.1.0"
