"""MI355X-native log-likelihood hot path of starry_process."""
__version__ = "0.1.0"
