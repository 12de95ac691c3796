"""Drop-in package path of the reference: ``from Demix.dNMF import ...`` (demo.py:8) resolves here."""
