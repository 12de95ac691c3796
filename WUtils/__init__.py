"""Drop-in package path of the reference's simulator helpers."""
