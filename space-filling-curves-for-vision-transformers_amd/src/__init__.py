"""Import-path aliases of the reference layout (`src.curves...`, `src.tokenizers...`, `src.models.vit`,
`src.training.train`): with this directory on sys.path the reference's own import lines resolve to
the MI355X implementation in `sfcvit`."""
