"""Mirror of the reference's `datasets` package for the multi-task pre-training path.
Only the batch scheduler is provided; the image datasets themselves (MultiGen-20M readers) are out of scope."""
