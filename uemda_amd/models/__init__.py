"""Mirror of the reference package `uemda.models` (import `uemda_amd.models.Encoder`)."""
