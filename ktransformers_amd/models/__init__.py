"""Only the model-side objects the hot path's operators touch (the paged latent KV cache)."""
