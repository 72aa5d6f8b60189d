# Convenience targets; the driver uses __graft_entry__.build()/smoke(), pytest and bench.py directly.
PY ?= python
.PHONY: build test test-gpu bench smoke clean
build:
	$(PY) -c "import __graft_entry__ as g; g.build()"
test: build
	$(PY) -m pytest tests -q -m "not gpu"
test-gpu: build
	$(PY) -m pytest tests -q -m gpu
smoke: build
	$(PY) -c "import __graft_entry__ as g; g.smoke()"
bench: build
	$(PY) bench.py
clean:
	$(MAKE) -C netobserv-ebpf-agent_amd/csrc clean
	$(MAKE) -C oracle clean
