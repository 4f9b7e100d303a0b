"""The routes a chunk of alignments can take through `classify.Engine`, one
module each (mixins of the engine; `Engine.run_chunk` dispatches)."""
