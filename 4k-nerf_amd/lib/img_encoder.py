"""Placeholder for the reference's `lib/img_encoder`.

`/root/reference/run.py:11` does `from lib import img_encoder, ...`, but the reference tree ships no `lib/img_encoder.py` (only a stale
`.pyc` in `lib/__pycache__/`) and `run.py` never touches the name again: the import alone keeps `run.py` from
starting.  With this package providing `lib`, the import resolves to this empty module, so `run.py` starts.
Nothing here is on the hot path and nothing is computed here."""
