"""Empty stand-in: reference modules/corr.py imports droid_backends at module scope -- GOLDEN GENERATION ONLY."""
