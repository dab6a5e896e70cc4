"""Application identity used to namespace run data (parity: alphatriangle/config/app_config.py:1)."""

APP_NAME: str = "AlphaTriangle"
