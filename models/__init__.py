"""Drop-in `models` package: the reference's import paths resolve to the MI355X implementation."""
