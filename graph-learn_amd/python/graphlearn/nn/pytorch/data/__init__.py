"""graphlearn.nn.pytorch.data: `Dataset` lives in dataset.py (import it from there, as the reference's examples do)."""
