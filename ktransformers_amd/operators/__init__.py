"""Host-side mirror of the reference's operator-injection surface (archive/ktransformers/operators)."""
