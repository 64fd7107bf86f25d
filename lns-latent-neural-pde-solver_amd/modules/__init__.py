"""Module names of the reference's `modules/` package that sit on the hot path."""
