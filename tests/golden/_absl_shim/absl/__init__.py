"""Minimal stand-in for absl-py, written for this repo (absl is not installed in the image).

Only used by tests/golden/make_golden.py so that the reference's config/config.py can be
imported unmodified when regenerating golden vectors.  Never imported by the product.
"""
