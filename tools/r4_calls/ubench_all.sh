#!/bin/bash
exec bash "$(dirname "$0")/../ubench/run_all.sh"
