"""nclone_amd -- MI355X-native batched N++ environment stepper (drop-in for nclone's step()/reset() hot path)."""
__version__ = "0.1.0"
