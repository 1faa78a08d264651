"""MI355X-native D2Q9 BGK lattice-Boltzmann path: a HIP shared library (csrc/) behind the Python
driver surface of cchan/latticeboltzmann.  See DESIGN.md."""
