"""Data layer of the hot path's callers: precomp feature files, vocabulary, tokenisers, the
reference's batch 8-tuple (itr/datamodule/*.py) and an asynchronous host->HBM feature stream."""
