"""Drop-in `utils` package: reference import paths resolve to the MI355X implementation."""
