"""QIIME 2 method surface: signature-compatible `classify`.

qiime2 itself is not required (and is absent from this environment): the
function takes plain Python inputs where the plugin takes artifact views.
"""
from .plugin import classify  # noqa: F401
