"""Task-type tags (same string values as pixelssl/utils/constant.py so plugins compare equal)."""
REGRESSION = 'regression'
CLASSIFICATION = 'classification'
