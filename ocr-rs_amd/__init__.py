"""MI355X-native OCR inference hot path (host-side mirror of lazareviczoran/ocr-rs).

The importable name of this package is ``ocr_rs_amd`` (see ../ocr_rs_amd.py); the
directory keeps the repository's ``ocr-rs_amd`` spelling.
"""
from . import weights  # noqa: F401
