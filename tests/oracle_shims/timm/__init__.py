"""Minimal stand-in for the `timm` helpers the reference imports (initialisers, DropPath, decorators).
None of these performs forward arithmetic when drop rates are 0 (the only case exercised)."""
