"""Spike encoders for the DCLL input (see utils.py)."""
